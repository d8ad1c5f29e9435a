"""File-mode sampling (didi_amd/hostio.py, the loop of scripts/sample.py:87-100) on the GPU with the tiny golden model."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(tmp_path, n=10):
    from golden_util import load_case, spec_to_config, weights_of
    from didi_amd import hostio
    from didi_amd.model import DiDiSampler
    from toy_tokenizer import toy_tokenizer

    _, meta = load_case("tiny_ddpm_f1")
    model = DiDiSampler(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), device="cuda:0", precision="bf16x3")
    tok = toy_tokenizer(meta["spec"]["vocab"], truncation_side="left")
    path = tmp_path / "test.jsonl"
    with open(path, "w") as f:
        for i in range(n):
            src = " ".join(f"w{(37 * i + 11 * j) % 590 + 5}" for j in range(2 + (i * 3) % 9))
            print(json.dumps({"src": src, "trg": f"w{i + 5}"}), file=f)
    samples = hostio.read_dialogs(str(path), hostio.Preprocessor(tok.cls_token + " ", " " + tok.sep_token))
    return hostio, model, tok, samples


def _run(hostio, model, tok, samples, out, **kw):
    torch.manual_seed(5)
    with hostio.JsonlWriter(str(out)) as w:
        n = hostio.sample_file(hostio.DialogBatches(samples, tok, 24, 4, **kw), model, "ddpm", 1, tok, w, target_len=23)
    assert n == len(samples)
    return [json.loads(x) for x in open(out)]


def test_file_mode_matches_sample_and_trimming_keeps_replies(tmp_path):
    from didi_amd.sampling import sample

    hostio, model, tok, samples = _setup(tmp_path)
    full = _run(hostio, model, tok, samples, tmp_path / "full.json")
    assert [r["source"] for r in full] == [s[0] for s in samples] and [r["reference"] for r in full] == [s[1] for s in samples]
    # the same batches through sample()'s own decode path (what scripts/sample.py:95-97 calls)
    torch.manual_seed(5)
    want = []
    for _, ctx, _, _ in hostio.DialogBatches(samples, tok, 24, 4):
        want += sample(ctx.to("cuda:0"), model, "ddpm", 1, tokenizer=tok, skip_special=False)
    assert [r["recover"] for r in full] == want
    # trimmed context padding (same batches, same noise): the replies do not change
    trimmed = _run(hostio, model, tok, samples, tmp_path / "trim.json", trim=8)
    a = [w for r in full for w in r["recover"].split()]
    b = [w for r in trimmed for w in r["recover"].split()]
    assert len(a) == len(b) and sum(x == y for x, y in zip(a, b)) >= 0.99 * len(a)
    # length-sorted batches change the batch composition (other noise), never the order or the bookkeeping of the file
    srt = _run(hostio, model, tok, samples, tmp_path / "sorted.json", trim=8, sort_by_length=True)
    assert [r["source"] for r in srt] == [s[0] for s in samples]
