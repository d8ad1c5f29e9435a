"""Lightning checkpoint loader: architecture inference from the reference's state-dict layout."""
import pytest
import torch

from golden_util import config1, config1_state_dict, load_case, spec_to_config, weights_of


def test_config_inferred_from_state_dict_shapes():
    from didi_amd.checkpoint import config_from_state_dict
    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import synthetic_state_dict

    cfg = config1()
    with pytest.raises(ValueError, match="dec_heads"):  # 128 wide: not a published configuration, heads must be given
        config_from_state_dict(config1_state_dict())
    got = config_from_state_dict(config1_state_dict(), dec_heads=4, enc_heads=4)
    for f in ("vocab", "dec_dim", "dec_layers", "dec_ffn", "enc_kind", "enc_dim", "enc_layers", "enc_ffn", "diffusion_steps", "tie_weights"):
        assert getattr(got, f) == getattr(cfg, f), f
    # head counts are not in the shapes: defaults follow get_components, overrides win
    assert config_from_state_dict(config1_state_dict(), dec_heads=4, enc_heads=4).dec_heads == 4
    big = DiDiConfig()  # blenderbot-400M-distill-sized encoder + BertConfig-default decoder
    sd = synthetic_state_dict(DiDiConfig(dec_layers=1, enc_layers=1, vocab=64, diffusion_steps=10), seed=1)
    got = config_from_state_dict(sd)
    assert (got.dec_heads, got.enc_heads, got.enc_dim, got.enc_ffn, got.has_adapter) == (big.dec_heads, big.enc_heads, 1280, 5120, True)
    assert got.diffusion_steps == 10 and got.dec_layers == 1


def test_bert_encoder_variant_and_tied_weights():
    from didi_amd.checkpoint import config_from_state_dict

    t, meta = load_case("tiny_bertenc_ddpm_f2")
    sd = weights_of(t)
    want = spec_to_config(meta["spec"])
    got = config_from_state_dict(sd, dec_heads=want.dec_heads, enc_heads=want.enc_heads)
    assert got.enc_kind == "bert" and got.has_adapter == want.has_adapter and got.enc_dim == want.enc_dim
    tied = dict(sd)
    tied["classifier.weight"] = tied["emb.weight"]
    assert config_from_state_dict(tied, dec_heads=want.dec_heads, enc_heads=want.enc_heads).tie_weights
    with pytest.raises(ValueError):
        config_from_state_dict({k: v for k, v in sd.items() if not k.startswith("encoder.")})  # no recognisable encoder
    with pytest.raises(ValueError):  # T5 is named, not built
        config_from_state_dict({**{k: v for k, v in sd.items() if not k.startswith("encoder.")},
                                "encoder.block.0.layer.0.SelfAttention.q.weight": torch.zeros(1)})


def test_load_state_reads_lightning_layout(tmp_path):
    from didi_amd.checkpoint import load_state

    sd = {k: v for k, v in list(config1_state_dict().items())[:5]}
    path = tmp_path / "step_10.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"diffusion_steps": 50, "schedule": "sqrt", "opt_kwargs": {"lr": 1e-4}},
                "epoch": 0, "global_step": 10}, path)
    got, hp = load_state(path)
    assert set(got) == set(sd) and hp["schedule"] == "sqrt"
    bare = tmp_path / "weights.pt"
    torch.save(sd, bare)
    got, hp = load_state(bare)
    assert set(got) == set(sd) and hp == {}


def test_t5_encoder_variants_inferred():
    """T5EncoderModel checkpoints (model.py:39-42): family, head count, d_kv, FFN width and the gated form come from shapes."""
    from golden_util import load_case, weights_of
    from didi_amd.checkpoint import config_from_state_dict

    for name, gated in (("tiny_t5enc_ddpm_f2", False), ("tiny_t5gated_euler_f3", True)):
        t, meta = load_case(name)
        spec = meta["spec"]
        cfg = config_from_state_dict(weights_of(t), dec_heads=spec["dec_heads"])
        assert cfg.enc_kind == "t5" and cfg.enc_gated_ffn == gated and cfg.enc_ln_eps == 1e-6
        assert (cfg.enc_dim, cfg.enc_heads, cfg.enc_dkv, cfg.enc_ffn, cfg.enc_layers, cfg.enc_rel_buckets) == \
            (spec["enc_dim"], spec["enc_heads"], spec["enc_dkv"], spec["enc_ffn"], spec["enc_layers"], 32)
        assert cfg.has_adapter == (spec["enc_dim"] != spec["dec_dim"])


class _FakeDictConfig(dict):
    """Stand-in for omegaconf.DictConfig (scripts/train.py:98 stores one as `opt_kwargs`): a class the weights-only loader rejects."""


def _decode(tokenizer, ids, skip_special_tokens=True):
    raise AssertionError("a checkpoint loader must never call into pickled callables")


def test_realistic_lightning_checkpoint_is_read_without_executing_anything(tmp_path):
    """hyper_parameters as the reference writes them (model.py:91, scripts/train.py:89-99): an OmegaConf-like mapping object and a
    functools.partial of a tokenizer method.  torch.load(weights_only=True) refuses the file; load_state returns the tensors and
    the plain hyper-parameters, with the foreign objects dropped -- and nothing of them imported or called."""
    import functools
    import pickle

    from didi_amd.checkpoint import config_from_state_dict, load_state

    cfg = config1()
    sd = config1_state_dict()
    tied = dict(sd)
    tied["classifier.weight"] = tied["emb.weight"]  # tie_weights: one storage, two keys (model.py:119-120)
    hp = {"diffusion_steps": cfg.diffusion_steps, "schedule": "sqrt", "step_freq": 5, "sampling_mode": "euler", "tie_weights": True,
          "opt_kwargs": _FakeDictConfig(lr=1e-4, scheduler=_FakeDictConfig(name="linear")),
          "batch_decoder": functools.partial(_decode, "tokenizer")}
    path = tmp_path / "step_100.ckpt"
    torch.save({"state_dict": tied, "hyper_parameters": hp, "epoch": 1, "global_step": 100,
                "optimizer_states": [{"state": {0: {"exp_avg": torch.ones(3)}}}], "callbacks": {"x": _FakeDictConfig(a=1)}}, path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    got, ghp = load_state(path)
    assert set(got) == set(tied) and all(torch.equal(got[k], tied[k]) for k in tied)
    assert ghp == {"diffusion_steps": cfg.diffusion_steps, "schedule": "sqrt", "step_freq": 5, "sampling_mode": "euler", "tie_weights": True}
    c = config_from_state_dict(got, dec_heads=4, enc_heads=4)
    assert c.tie_weights and c.vocab == cfg.vocab
    # adapter-absent variant (enc_dim == dec_dim, e.g. bert-base contexts): same path
    t, meta = load_case("tiny_bertenc_ddpm_f2")
    w = weights_of(t)
    p2 = tmp_path / "bert.ckpt"
    torch.save({"state_dict": w, "hyper_parameters": {"batch_decoder": functools.partial(_decode, 1)}}, p2)
    got2, hp2 = load_state(p2)
    assert hp2 == {} and all(torch.equal(got2[k], w[k]) for k in w)
    want = spec_to_config(meta["spec"])
    assert config_from_state_dict(got2, dec_heads=want.dec_heads, enc_heads=want.enc_heads).has_adapter == want.has_adapter


def test_tensor_only_reader_never_imports_foreign_globals(tmp_path):
    """A pickle that names an importable module with side effects (os.system) must come back inert."""
    import io
    import pickle
    import zipfile

    from didi_amd.checkpoint import _Inert, _read_tensors_only

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ("touch " + str(tmp_path / "pwned"),))

    buf = io.BytesIO()
    pickle.dump({"state_dict": {}, "hyper_parameters": {"x": Evil(), "n": 3}}, buf, protocol=2)
    path = tmp_path / "evil.ckpt"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("archive/data.pkl", buf.getvalue())
    tree, stubbed = _read_tensors_only(path)
    assert not (tmp_path / "pwned").exists()
    assert isinstance(tree["hyper_parameters"]["x"], _Inert) and tree["hyper_parameters"]["n"] == 3
    assert any(name.endswith(".system") for name in stubbed)
