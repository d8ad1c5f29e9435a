"""Host-side I/O around the sampler (didi_amd/hostio.py) against the reference behaviour it restates:
src/data/utils.py:6-23 (Preprocessor), src/data/commonsense_dataset.py:38-42,72-76 (file format, test_collate_fn),
scripts/sample.py:90-100 (output file name and lines).  Tokenizers are built offline (tests/toy_tokenizer.py)."""
import json

import pytest
import torch

from didi_amd import hostio
from toy_tokenizer import toy_tokenizer


def _dialogs(tmp_path, n=11):
    path = tmp_path / "test.jsonl"
    rows = []
    with open(path, "w") as f:
        for i in range(n):
            src = [" ".join(f"w{(7 * i + j) % 50 + 5}" for j in range(1 + (i * 5) % 13)) for _ in range(1 + i % 3)]
            row = {"src": src if i % 2 else src[0], "trg": f"w{i + 5} w{i + 6}"}
            rows.append(row)
            print(json.dumps(row), file=f)
        print("", file=f)  # trailing blank line is tolerated
    return str(path), rows


def test_preprocessor_matches_reference_rules():
    tok = toy_tokenizer(64)
    p = hostio.Preprocessor.for_tokenizer(tok, "bert")
    assert (p.bos, p.eos, p.sep) == ("[CLS]", "[SEP]", "[CLS] [SEP]")
    assert p(["a", "b"], "c") == ("[CLS]a[CLS] [SEP]b[SEP]", "[CLS]c[SEP]")
    assert p("a", "c") == ("[CLS]a[SEP]", "[CLS]c[SEP]")
    b = hostio.Preprocessor.for_tokenizer(tok, "blenderbot")
    assert (b.bos, b.eos) == (tok.bos_token, tok.eos_token)
    t5 = hostio.Preprocessor.for_tokenizer(tok, "t5")
    assert t5(["a", "b"], "c") == ("a\nb" + tok.eos_token, "c" + tok.eos_token)
    with pytest.raises(ValueError):
        hostio.Preprocessor.for_tokenizer(tok, "gpt")


def test_default_batches_are_test_collate_fn(tmp_path):
    tok = toy_tokenizer(64, truncation_side="left")
    path, rows = _dialogs(tmp_path)
    pre = hostio.Preprocessor.for_tokenizer(tok, "bert")
    samples = hostio.read_dialogs(path, pre)
    assert len(samples) == len(rows) and samples[1] == pre(rows[1]["src"], rows[1]["trg"])
    batches = list(hostio.DialogBatches(samples, tok, max_context_len=16, batch_size=4))
    assert [len(b[0]) for b in batches] == [4, 4, 3] and sum((b[0] for b in batches), []) == list(range(11))
    index, ctx, str_ctx, str_rep = batches[1]
    want = tok(list(str_ctx), max_length=16, padding="max_length", truncation=True, return_tensors="pt", add_special_tokens=False)
    assert torch.equal(ctx.input_ids, want.input_ids) and torch.equal(ctx.attention_mask, want.attention_mask)
    assert ctx.input_ids.shape == (4, 16) and ctx.input_ids.dtype == torch.int64
    assert str_ctx == tuple(samples[i][0] for i in index) and str_rep == tuple(samples[i][1] for i in index)
    # left truncation keeps the END of a long context (the latest utterances)
    long = [("[CLS] " + " ".join(f"w{5 + i}" for i in range(40)) + " [SEP]", "[CLS] w5 [SEP]")]
    _, c, _, _ = next(iter(hostio.DialogBatches(long, tok, 16, 1)))
    assert c.input_ids[0, -1].item() == tok.sep_token_id and int(c.attention_mask.sum()) == 16


def test_trim_and_length_sort(tmp_path):
    tok = toy_tokenizer(64, truncation_side="left")
    path, _ = _dialogs(tmp_path, 23)
    samples = hostio.read_dialogs(path, hostio.Preprocessor.for_tokenizer(tok, "bert"))
    full = {i: (ids, m) for idx, c, _, _ in hostio.DialogBatches(samples, tok, 48, 5) for i, ids, m in zip(idx, c.input_ids, c.attention_mask)}
    seen, widths = [], []
    for idx, c, _, _ in hostio.DialogBatches(samples, tok, 48, 5, trim=8, sort_by_length=True):
        w = c.input_ids.shape[1]
        widths.append(w)
        assert w % 8 == 0 and w <= 48 and int(c.attention_mask.sum(1).max()) > w - 8
        for i, ids, m in zip(idx, c.input_ids, c.attention_mask):
            assert torch.equal(ids, full[i][0][:w]) and torch.equal(m, full[i][1][:w]) and int(full[i][1][w:].sum()) == 0
        seen += idx
    assert sorted(seen) == list(range(23)) and seen != list(range(23)) and widths == sorted(widths) and widths[0] < widths[-1]
    with pytest.raises(ValueError):
        hostio.trim_padding(torch.ones(2, 4, dtype=torch.int64), torch.tensor([[0, 1, 1, 1], [1, 1, 0, 0]]))
    ids, m = hostio.trim_padding(torch.zeros(2, 32, dtype=torch.int64), torch.zeros(2, 32, dtype=torch.int64), 16)
    assert ids.shape == (2, 16)  # all-padding batch keeps one granule
    with pytest.raises(ValueError):
        hostio.DialogBatches(samples, tok, 48, 5, trim=0)


def _fake_sample(eos):
    """ids[b] = the context's first tokens reversed, then eos, then junk: checks truncation and the call arguments."""
    calls = []

    def fn(raw_context, model, mode, step_freq, tokenizer=None, return_ids=False, target_len=None, **kw):
        assert return_ids and tokenizer is not None
        calls.append((tuple(raw_context.input_ids.shape), mode, step_freq, target_len))
        ids = raw_context.input_ids[:, :3].flip(1)
        S = target_len or raw_context.input_ids.shape[1] - 1
        out = torch.full((ids.shape[0], S), 9, dtype=torch.int64)
        out[:, :3], out[:, 3] = ids, eos
        return out, torch.full((ids.shape[0],), 4, dtype=torch.int32)
    return fn, calls


def test_sample_file_writes_reference_lines_in_file_order(tmp_path):
    tok = toy_tokenizer(64, truncation_side="left")
    path, _ = _dialogs(tmp_path, 23)
    samples = hostio.read_dialogs(path, hostio.Preprocessor.for_tokenizer(tok, "bert"))
    fn, calls = _fake_sample(tok.sep_token_id)
    out = tmp_path / hostio.JsonlWriter.file_name(3, "ddpm", 10)
    assert out.name == "seed3_ddpm_freq10.json"
    with hostio.JsonlWriter(str(out)) as w:
        n = hostio.sample_file(hostio.DialogBatches(samples, tok, 48, 5, trim=8, sort_by_length=True), None, "ddpm", 10, tok, w,
                               target_len=47, sample_fn=fn, device="cpu")
    assert n == 23 and len(calls) == 5 and all(c[1:] == ("ddpm", 10, 47) for c in calls)
    lines = [json.loads(x) for x in open(out)]
    assert len(lines) == 23
    for i, rec in enumerate(lines):
        assert list(rec) == ["recover", "reference", "source"]
        assert rec["source"] == samples[i][0] and rec["reference"] == samples[i][1]
        first3 = tok([samples[i][0]], add_special_tokens=False).input_ids[0][:3][::-1]
        assert rec["recover"] == tok.decode(first3 + [tok.sep_token_id], skip_special_tokens=False)


def test_sample_file_surfaces_worker_and_sampler_errors(tmp_path):
    tok = toy_tokenizer(64)
    samples = [("[CLS] w5 [SEP]", "[CLS] w6 [SEP]")] * 6

    class BadTok:
        sep_token_id = tok.sep_token_id

        def batch_decode(self, *a, **k):
            raise RuntimeError("decode failed")

    fn, _ = _fake_sample(tok.sep_token_id)
    w = hostio.JsonlWriter(str(tmp_path / "o.json"))
    with pytest.raises(RuntimeError, match="decode failed"):
        hostio.sample_file(hostio.DialogBatches(samples, tok, 8, 2), None, "ddpm", 1, BadTok(), w, sample_fn=fn, device="cpu")

    def boom(*a, **k):
        raise NotImplementedError("No x sampling strategy")

    with pytest.raises(NotImplementedError):
        hostio.sample_file(hostio.DialogBatches(samples, tok, 8, 2), None, "x", 1, tok, w, sample_fn=boom, device="cpu")
    w.put(5, "a", "b", "c")  # a hole before record 5
    with pytest.raises(RuntimeError, match="never written"):
        w.close()


def test_preprocessor_against_the_reference_class():
    """tests/golden/preprocessor.json holds what the REFERENCE's ``Preprocessor`` (src/data/utils.py:6-23, executed from the
    reference file by oracle/make_golden.py preproc) returned for the three tokenizer families: the same special-token strings
    in, the same (context, reply) strings out -- utterance lists joined by the family's separator, empty strings, an empty
    list -- and ``get_mode``'s error for a name of no family."""
    import json
    import os

    import pytest

    from didi_amd import hostio

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocessor.json")))
    assert len(fx["families"]) == 3

    class Tok:
        pass

    for fam in fx["families"]:
        tok = Tok()
        tok.__dict__.update(fam["tokens"])
        assert hostio.family_of(fam["base_name"]) == fam["family"]
        for p in (hostio.Preprocessor.for_tokenizer(tok, fam["family"]), hostio.Preprocessor.for_tokenizer(tok, fam["base_name"])):
            assert (p.bos, p.eos, p.sep) == (fam["bos"], fam["eos"], fam["sep"])
            for c in fam["cases"]:
                assert list(p(c["src"], c["trg"])) == c["out"], (fam["family"], c)
    with pytest.raises(ValueError) as e:
        hostio.family_of(fx["unsupported"]["base_name"])
    assert str(e.value) == fx["unsupported"]["error"]
