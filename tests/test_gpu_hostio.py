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


def test_file_mode_reproduces_the_reference_fixture(tmp_path):
    """SURVEY 8f row 4 against the REFERENCE's outputs: the contexts of fixture tiny_ddpm_f1 written out as a dialog file, read
    back through read_dialogs -> DialogBatches (the tokenizer call of CommonSenseDataset.test_collate_fn,
    commonsense_dataset.py:72-76) -> sample_file -> JsonlWriter (scripts/sample.py:87-100), with the fixture's noise.  The token
    ids the tokenizer produces must be the fixture's input ids, and the replies in the file, tokenised again, must be the ids the
    reference sampler produced (out.ids of the imported src.sampling.sample run), each row cut after its first eos as
    truncate_predictions does (src/sampling.py:104-113)."""
    from golden_util import load_case, noise_list, spec_to_config, weights_of
    from didi_amd import hostio
    from didi_amd.model import DiDiSampler
    from toy_tokenizer import SPECIALS, toy_tokenizer

    t, meta = load_case("tiny_ddpm_f1")
    run = meta["run"]
    V = meta["spec"]["vocab"]
    tok = toy_tokenizer(V, truncation_side="left")
    word = lambda i: SPECIALS[i] if i < len(SPECIALS) else f"w{i}"
    ids_in, mask = t["in.input_ids"], t["in.attention_mask"]
    B, C = ids_in.shape
    path = tmp_path / "test.jsonl"
    with open(path, "w") as f:
        for b in range(B):
            print(json.dumps({"src": " ".join(word(int(i)) for i in ids_in[b, : int(mask[b].sum())]), "trg": f"w{b + 5}"}), file=f)
    samples = hostio.read_dialogs(str(path), hostio.Preprocessor("", ""))
    batches = hostio.DialogBatches(samples, tok, C, B)
    (_, ctx, _, _), = list(batches)
    assert torch.equal(ctx.input_ids, ids_in) and torch.equal(ctx.attention_mask, mask)
    model = DiDiSampler(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), device="cuda:0", precision="bf16x3")
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    out = tmp_path / hostio.JsonlWriter.file_name(0, run["mode"], run["step_freq"])
    with hostio.JsonlWriter(str(out)) as w:
        assert hostio.sample_file(batches, model, run["mode"], run["step_freq"], tok, w, noise=noise) == B
    rows = [json.loads(x) for x in open(out)]
    assert [r["source"] for r in rows] == [s[0] for s in samples] and [r["reference"] for r in rows] == [s[1] for s in samples]
    back = {word(i): i for i in range(V)}
    eos = tok.eos_token_id or tok.sep_token_id
    for b, r in enumerate(rows):
        want = t["out.ids"][b].tolist()
        if eos in want:
            want = want[: want.index(eos) + 1]
        assert [back[x] for x in r["recover"].split()] == want, f"dialog {b}"
