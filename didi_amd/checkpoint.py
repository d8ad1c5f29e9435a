"""Lightning checkpoint -> DiDiSampler.

The reference saves with ``ModelCheckpoint`` (src/training.py:27-33) and reloads with
``DiDi.load_from_checkpoint(path, enc_dim, dec_dim, map_location)`` (scripts/sample.py:58).  A
``.ckpt`` is a dict with ``state_dict`` (key layout: SURVEY.md section 5) and ``hyper_parameters``
(``save_hyperparameters(ignore=[encoder, decoder])``, src/diffusion/model.py:91).  The shapes in the
state dict determine the architecture; what shapes cannot tell (attention head counts, LayerNorm
eps) follows the reference's ``get_components`` choices (model.py:29-54) and can be overridden.

Nothing in a checkpoint is ever executed.  ``torch.load(..., weights_only=True)`` is tried first; a real reference checkpoint
makes it refuse, because ``save_hyperparameters`` stored ``opt_kwargs`` (an OmegaConf ``DictConfig``, scripts/train.py:98) and
``batch_decoder`` (a ``functools.partial`` of a tokenizer method, scripts/train.py:89,99).  Such files go through
``_read_tensors_only``: an unpickler whose ``find_class`` resolves ONLY the handful of names a tensor record needs
(``collections.OrderedDict``, ``torch._utils._rebuild_tensor_v2`` / ``_rebuild_parameter``, the ``torch.*Storage`` tags) and
answers every other global with an inert stub that is never imported and never called into -- ``DictConfig``, ``partial`` and
whatever else the file names come back as empty placeholders, the tensors and plain scalars as themselves.
"""
import collections
import io
import pickle
import re
import zipfile

import torch

from .config import DiDiConfig

# hyper_parameters the sampler reads (DiDi.__init__ keyword names -> DiDiConfig fields)
_HPARAMS = {
    "diffusion_steps": "diffusion_steps", "schedule": "schedule", "step_freq": "step_freq", "pad_idx": "pad_idx",
    "noise_factor": "noise_factor", "tie_weights": "tie_weights", "sampling_mode": "sampling_mode",
    "s_churn": "s_churn", "s_tmin": "s_tmin", "s_tmax": "s_tmax",
}


# published head counts by model width (what `AutoModel.from_pretrained(name)` / `BertConfig()` give the reference, model.py:29-54);
# anything else must be passed explicitly: head counts are not recoverable from tensor shapes
_DEC_HEADS_DEFAULT = 12  # BertConfig.num_attention_heads, whatever hidden_size is (configs/train.yaml overrides only the dropouts)
_BLENDERBOT_HEADS = {512: 16, 1280: 32, 2560: 32}   # blenderbot_small-90M, blenderbot-400M-distill, blenderbot-3B
_BERT_HEADS = {128: 2, 256: 4, 512: 8, 768: 12, 1024: 16}  # google/bert_uncased_L-*_H-*_A-* family, bert-base, bert-large


def _n_layers(sd, pattern):
    idx = {int(m.group(1)) for k in sd for m in [re.match(pattern, k)] if m}
    return max(idx) + 1 if idx else 0


def config_from_state_dict(sd, **overrides) -> DiDiConfig:
    """Infer the DiDiConfig of a reference state dict from its tensor shapes."""
    c = {}
    c["vocab"], c["dec_dim"] = sd["emb.weight"].shape
    c["diffusion_steps"] = sd["time_embeds.weight"].shape[0] - 1
    c["dec_layers"] = _n_layers(sd, r"decoder\.encoder\.layer\.(\d+)\.")
    c["dec_ffn"] = sd["decoder.encoder.layer.0.intermediate.dense.weight"].shape[0]
    c["dec_max_pos"] = sd["decoder.embeddings.position_embeddings.weight"].shape[0]
    c["dec_heads"] = _DEC_HEADS_DEFAULT if c["dec_dim"] % (8 * _DEC_HEADS_DEFAULT) == 0 else None
    if "encoder.layers.0.self_attn.q_proj.weight" in sd:  # BlenderbotEncoder (model.py:31-33)
        c["enc_kind"] = "blenderbot"
        c["enc_dim"] = sd["encoder.embed_tokens.weight"].shape[1]
        c["enc_layers"] = _n_layers(sd, r"encoder\.layers\.(\d+)\.")
        c["enc_ffn"] = sd["encoder.layers.0.fc1.weight"].shape[0]
        c["enc_max_pos"] = sd["encoder.embed_positions.weight"].shape[0]
        c["enc_heads"] = _BLENDERBOT_HEADS.get(c["enc_dim"])
        c["enc_ln_eps"] = 1e-5
    elif "encoder.encoder.layer.0.attention.self.query.weight" in sd:  # BertModel (model.py:36-38)
        c["enc_kind"] = "bert"
        c["enc_dim"] = sd["encoder.embeddings.word_embeddings.weight"].shape[1]
        c["enc_layers"] = _n_layers(sd, r"encoder\.encoder\.layer\.(\d+)\.")
        c["enc_ffn"] = sd["encoder.encoder.layer.0.intermediate.dense.weight"].shape[0]
        c["enc_max_pos"] = sd["encoder.embeddings.position_embeddings.weight"].shape[0]
        c["enc_heads"] = _BERT_HEADS.get(c["enc_dim"])
        c["enc_ln_eps"] = 1e-12
    elif "encoder.encoder.block.0.layer.0.SelfAttention.q.weight" in sd:  # T5EncoderModel (model.py:39-42)
        q = "encoder.encoder.block.0.layer."
        rel = sd[q + "0.SelfAttention.relative_attention_bias.weight"]  # [buckets, heads]
        inner, c["enc_dim"] = sd[q + "0.SelfAttention.q.weight"].shape
        c["enc_kind"] = "t5"
        c["enc_layers"] = _n_layers(sd, r"encoder\.encoder\.block\.(\d+)\.")
        c["enc_rel_buckets"], c["enc_heads"] = rel.shape
        c["enc_dkv"] = inner // c["enc_heads"]
        c["enc_ffn"] = sd[q + "1.DenseReluDense.wo.weight"].shape[1]
        c["enc_gated_ffn"] = q + "1.DenseReluDense.wi_0.weight" in sd
        c["enc_ln_eps"] = 1e-6
    else:
        raise ValueError("unrecognised context encoder in the state dict")
    has_adapter = "adapter.0.weight" in sd
    if has_adapter != (c["enc_dim"] != c["dec_dim"]):
        raise ValueError("adapter weights and encoder/decoder widths disagree (model.py:122)")
    cls = sd.get("classifier.weight")
    c["tie_weights"] = cls is None or cls.data_ptr() == sd["emb.weight"].data_ptr() or torch.equal(cls, sd["emb.weight"])
    c.update(overrides)
    for side in ("dec", "enc"):
        heads, dim = c[side + "_heads"], c[side + "_dim"]
        if heads is None:
            raise ValueError(f"{side}_heads cannot be inferred from the checkpoint ({side}_dim = {dim} is not a published configuration): "
                             f"pass {side}_heads= explicitly")
        if c.get("enc_kind") == "t5" and side == "enc":
            continue  # T5: heads x d_kv need not equal d_model
        if dim % heads != 0:
            raise ValueError(f"{side}_heads = {heads} does not divide {side}_dim = {dim}")
    return DiDiConfig(**c)


class _Inert:
    """Placeholder for every global of a checkpoint that is not part of a tensor record.  It is what the unpickler gets INSTEAD of
    importing the named class or function: constructing, calling, subscripting or setting state on it does nothing."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __reduce__(self):  # never re-pickled with content
        return (_Inert, ())

    def __setitem__(self, k, v):
        pass

    def __getitem__(self, k):
        return _Inert()

    def append(self, x):
        pass

    def extend(self, xs):
        pass

    def __repr__(self):
        return "<inert placeholder>"


_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16,
    "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16, "CharStorage": torch.int8,
    "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class _StorageTag:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor(storage, storage_offset, size, stride, *unused, **unused_kw):
    """torch._utils._rebuild_tensor_v2 restated on a flat 1-D tensor (the storage): pure indexing."""
    if not isinstance(storage, torch.Tensor):
        return _Inert()
    return torch.as_strided(storage, tuple(size), tuple(stride), storage_offset).clone()


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *unused):
    return data


class _TensorOnlyUnpickler(pickle.Unpickler):
    """Resolves only what a tensor record needs; every other global becomes ``_Inert`` WITHOUT being imported."""

    def __init__(self, file, read_record):
        super().__init__(file)
        self._read_record = read_record
        self.stubbed = set()

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
            return _StorageTag(_STORAGE_DTYPES[name])
        self.stubbed.add(f"{module}.{name}")
        return _Inert

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel): the bytes live in the archive's data/<key> record
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageTag)):
            return _Inert()
        data = self._read_record(str(pid[2]))
        return torch.frombuffer(bytearray(data), dtype=pid[1].dtype) if len(data) else torch.empty(0, dtype=pid[1].dtype)


def _read_tensors_only(path):
    """A torch zip checkpoint -> its object tree with tensors and plain Python values intact and everything else inert."""
    with zipfile.ZipFile(path) as z:
        pkl = [n for n in z.namelist() if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a torch zip checkpoint (no single data.pkl record)")
        prefix = pkl[0][: -len("data.pkl")]
        if (prefix + "byteorder") in z.namelist() and z.read(prefix + "byteorder").strip() not in (b"little", b""):
            raise ValueError(f"{path}: big-endian checkpoint")
        up = _TensorOnlyUnpickler(io.BytesIO(z.read(pkl[0])), lambda key: z.read(f"{prefix}data/{key}"))
        return up.load(), sorted(up.stubbed)


def load_state(path):
    """(state_dict, hyper_parameters) of a Lightning ``.ckpt`` or of a bare state-dict file.  Placeholders (``_Inert``) stand in
    for hyper-parameters that are not plain data; they are dropped from the returned dict."""
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        ckpt, _ = _read_tensors_only(path)  # a real reference checkpoint: DictConfig / functools.partial in hyper_parameters
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        hp = ckpt.get("hyper_parameters")
        hp = {k: v for k, v in hp.items() if not isinstance(v, _Inert)} if isinstance(hp, dict) else {}
        sd = ckpt["state_dict"]
    else:
        sd, hp = ckpt, {}
    bad = [k for k, v in sd.items() if not isinstance(v, torch.Tensor)]
    if bad:
        raise ValueError(f"{path}: state_dict entries that are not tensors: {bad[:5]}")
    return sd, hp


def load_lightning_checkpoint(path, device="cuda", precision="fp16-sched", **overrides):
    """``DiDi.load_from_checkpoint`` for the sampler path -> DiDiSampler."""
    from .model import DiDiSampler

    sd, hp = load_state(path)
    known = {field: hp[name] for name, field in _HPARAMS.items() if name in hp and isinstance(hp[name], (int, float, str, bool))}
    if "diffusion_steps" in known and known["diffusion_steps"] + 1 != sd["time_embeds.weight"].shape[0]:
        raise ValueError("hyper_parameters.diffusion_steps does not match time_embeds.weight")
    known.update(overrides)
    cfg = config_from_state_dict(sd, **known)
    return DiDiSampler(sd, cfg, device=device, precision=precision)
