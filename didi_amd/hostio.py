"""Host-side I/O around the sampler: dialog files in, batches through
``sample``, ``{"recover", "reference", "source"}`` lines out.

Mirrors what ``scripts/sample.py:77-100`` does with ``CommonSenseDataset`` (src/data/commonsense_dataset.py)
and ``Preprocessor`` (src/data/utils.py:6-23), minus everything that needs the network: tokenizers are
passed in as objects (any HF-style tokenizer: callable with ``max_length/padding/truncation/return_tensors/
add_special_tokens``, plus ``batch_decode``), never fetched by name.

Two things differ from the reference, both optional and both off by default:
  * ``trim``: the reference pads every context to ``max_context_len`` (``padding="max_length"``,
    commonsense_dataset.py:27-32).  Padded keys carry zero attention weight, so the columns that are
    padding in EVERY row of a batch can be dropped without changing a valid position; the reply length
    stays ``max_context_len - 1`` through ``sample(..., target_len=)``.  ``sort_by_length`` groups dialogs of
    similar length so that the trimmed batches are short; the writer restores file order.
  * the decode of batch i (D2H copy, eos truncation, ``batch_decode``, json) runs on a worker thread
    while the GPU samples batch i+1; the reference does them in sequence (scripts/sample.py:95-100).
"""
import json
import queue
import threading
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch


FAMILIES = ("bert", "blenderbot", "t5")  # the order of the reference's `Modes` enum: the first name found in the base name wins


def family_of(base_name: str) -> str:
    """``get_mode`` (src/diffusion/model.py:22-26): the tokenizer / encoder family a hub model name belongs to."""
    for fam in FAMILIES:
        if fam in base_name:
            return fam
    raise ValueError(f"Unsupported base name: {base_name}")


class Preprocessor:
    """src/data/utils.py:6-23: join the utterances of a dialog and wrap context and reply in bos/eos."""

    def __init__(self, bos: str, eos: str, sep: Optional[str] = None):
        self.bos, self.eos = bos, eos
        self.sep = f"{bos} {eos}" if sep is None else sep

    @classmethod
    def for_tokenizer(cls, tokenizer, family: str):
        """``family``: "bert" (cls/sep), "blenderbot" (bos/eos) or "t5" ("" / newline / eos), as utils.py:11-18; a hub model
        name is accepted too (``family_of``)."""
        if family not in FAMILIES:
            family = family_of(family)
        if family == "bert":
            return cls(tokenizer.cls_token, tokenizer.sep_token)
        if family == "blenderbot":
            return cls(tokenizer.bos_token, tokenizer.eos_token)
        if family == "t5":
            return cls("", tokenizer.eos_token, sep="\n")
        raise ValueError(f"unknown tokenizer family {family!r}")

    def __call__(self, src, trg) -> Tuple[str, str]:
        if isinstance(src, list):
            src = self.sep.join(src)
        return self.bos + src + self.eos, self.bos + trg + self.eos


def read_dialogs(path: str, preprocess: Callable) -> List[Tuple[str, str]]:
    """One json object per line with "src" (utterance or list of utterances) and "trg" (commonsense_dataset.py:38-42)."""
    samples = []
    with open(path, "rt") as f:
        for line in f:
            if line.strip():
                d = json.loads(line)
                samples.append(preprocess(d["src"], d["trg"]))
    return samples


class ContextBatch:
    """What ``sample`` reads from a HF BatchEncoding: ``.input_ids`` / ``.attention_mask`` [B, C] int64."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device, non_blocking=False):
        return ContextBatch(self.input_ids.to(device, non_blocking=non_blocking), self.attention_mask.to(device, non_blocking=non_blocking))


class DialogBatches:
    """Iterates ``(index, raw_context, str_contexts, str_replies)`` over a list of preprocessed dialogs.

    With the defaults each item is ``CommonSenseDataset.test_collate_fn`` (commonsense_dataset.py:72-76) of
    ``batch_size`` consecutive dialogs: contexts tokenised with ``padding="max_length"``, ``truncation=True``,
    ``add_special_tokens=False`` to ``max_context_len`` (the tokenizer's own ``truncation_side`` decides which end
    is cut; the reference builds it with "left").  ``index`` lists the dialogs' positions in ``samples``.
    """

    def __init__(self, samples: Sequence[Tuple[str, str]], context_tokenizer, max_context_len: int, batch_size: int,
                 trim: Optional[int] = None, sort_by_length: bool = False):
        if trim is not None and trim < 1:
            raise ValueError("trim must be a positive column granule or None")
        self.samples, self.tok = samples, context_tokenizer
        self.max_context_len, self.batch_size, self.trim = max_context_len, batch_size, trim
        self.order = list(range(len(samples)))
        if sort_by_length:
            n_tok = self._lengths()
            self.order.sort(key=lambda i: (n_tok[i], i))

    def _encode(self, texts):
        return self.tok(list(texts), max_length=self.max_context_len, padding="max_length", truncation=True,
                        return_tensors="pt", add_special_tokens=False)

    def _lengths(self):
        out = []
        for a in range(0, len(self.samples), 1024):
            enc = self._encode([s[0] for s in self.samples[a:a + 1024]])
            out.extend(enc.attention_mask.sum(1).tolist())
        return out

    def __len__(self):
        return -(-len(self.samples) // self.batch_size)

    def __iter__(self) -> Iterator[Tuple[List[int], ContextBatch, Tuple[str, ...], Tuple[str, ...]]]:
        for a in range(0, len(self.order), self.batch_size):
            index = self.order[a:a + self.batch_size]
            str_contexts, str_replies = zip(*(self.samples[i] for i in index))
            enc = self._encode(str_contexts)
            ids, mask = enc.input_ids, enc.attention_mask
            if self.trim is not None:
                ids, mask = trim_padding(ids, mask, self.trim)
            yield index, ContextBatch(ids, mask), str_contexts, str_replies


def trim_padding(input_ids: torch.Tensor, attention_mask: torch.Tensor, granule: int = 16):
    """Drop the trailing columns that are padding in every row, keeping a multiple of ``granule`` columns
    (at least one granule, never more than there were).  Rows must be left-aligned (right padding)."""
    m = attention_mask != 0
    if bool((m[:, 1:] & ~m[:, :-1]).any()):
        raise ValueError("contexts are not right-padded")
    n = max(int(m.sum(1).max().item()), 1)
    keep = min(input_ids.shape[1], -(-n // granule) * granule)
    return input_ids[:, :keep].contiguous(), attention_mask[:, :keep].contiguous()


class JsonlWriter:
    """``print(json.dumps({"recover", "reference", "source"}), file=f)`` of scripts/sample.py:99-100, in dialog order:
    records may arrive out of order (length-sorted batches) and are held back until their predecessors are written."""

    def __init__(self, path: str):
        self.f = open(path, "w")
        self.next, self.pending = 0, {}

    @staticmethod
    def file_name(seed: int, mode: str, freq: int) -> str:
        return f"seed{seed}_{mode}_freq{freq}.json"  # scripts/sample.py:90

    def put(self, index: int, recover: str, reference: str, source: str):
        self.pending[index] = json.dumps({"recover": recover, "reference": reference, "source": source})
        while self.next in self.pending:
            print(self.pending.pop(self.next), file=self.f)
            self.next += 1

    def close(self):
        if self.pending:
            missing = self.next
            self.f.close()
            raise RuntimeError(f"dialog {missing} was never written ({len(self.pending)} later records held back)")
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:
            self.f.close()


def decode_replies(ids_host: torch.Tensor, lengths: Sequence[int], tokenizer, skip_special: bool) -> List[str]:
    """src/sampling.py:34-35 after the device-side eos scan: cut each row at its length, ``batch_decode``."""
    return tokenizer.batch_decode([ids_host[i, : int(lengths[i])] for i in range(ids_host.shape[0])], skip_special_tokens=skip_special)


def sample_file(batches: Iterable, model, mode: str, step_freq: int, reply_tokenizer, writer: JsonlWriter, *,
                target_len: Optional[int] = None, skip_special: bool = False, device=None, sample_fn=None, **sample_kw) -> int:
    """The loop of scripts/sample.py:91-100 with decode and write overlapped with the next batch's sampling.

    ``batches`` yields ``(index, raw_context, str_contexts, str_replies)`` (DialogBatches).  ``target_len`` pins the reply
    length when contexts are trimmed (pass ``max_context_len - 1`` to reproduce the untrimmed shapes).  ``skip_special``
    defaults to False as the reference's file mode passes it (scripts/sample.py:96).  Returns the number of dialogs written.
    """
    if sample_fn is None:
        from .sampling import sample as sample_fn
    device = device if device is not None else getattr(model, "device", "cpu")
    on_gpu = torch.device(device).type == "cuda"
    work: "queue.Queue" = queue.Queue(maxsize=2)  # at most two decoded-later batches in flight
    failure: List[BaseException] = []

    def worker():
        while True:
            item = work.get()
            if item is None:
                return
            if failure:
                continue  # keep draining so the producer never blocks
            try:
                index, ids_host, len_host, ready, str_contexts, str_replies = item
                if ready is not None:
                    ready.synchronize()
                replies = decode_replies(ids_host, len_host.tolist(), reply_tokenizer, skip_special)
                for i, recov, ref, src in zip(index, replies, str_replies, str_contexts):
                    writer.put(i, recov, ref, src)
            except BaseException as e:  # surfaced by the producer below
                failure.append(e)

    th = threading.Thread(target=worker, name="didi-decode", daemon=True)
    th.start()
    n = 0
    try:
        for index, raw_context, str_contexts, str_replies in batches:
            if failure:
                break
            raw_context = raw_context.to(device)
            ids, lengths = sample_fn(raw_context, model, mode, step_freq, tokenizer=reply_tokenizer, return_ids=True,
                                     **({"target_len": target_len} if target_len is not None else {}), **sample_kw)
            if on_gpu:
                ids_host = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
                len_host = torch.empty(lengths.shape, dtype=lengths.dtype, pin_memory=True)
                ids_host.copy_(ids, non_blocking=True)
                len_host.copy_(lengths, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
            else:
                ids_host, len_host, ready = ids, lengths, None
            work.put((index, ids_host, len_host, ready, str_contexts, str_replies))
            n += len(index)
    finally:
        work.put(None)
        th.join()
    if failure:
        raise failure[0]
    return n
