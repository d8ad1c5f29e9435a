"""``DiDi.training_step`` + ``configure_optimizers`` on the HIP path: forward, backward, gradient all-reduce, AdamW.

Reference: src/diffusion/model.py:200-243 (training_step), :136-158 (AdamW(lr=1) x LambdaLR), :271-308 (the two LR lambdas),
configs/train.yaml:33-45 (optimizer block, gradient_clip_val), src/training.py:37 (DDP: gradient all-reduce, mean).
The backward is written out by hand (no autograd): every piece is a libdidi_hip kernel --
  * Linear: dgrad and wgrad are MFMA GEMMs on transposed split planes, the bias gradient a column sum (ops.linear_backward);
  * LayerNorm, GELU, tanh, softmax cross-entropy, embedding scatter-add, attention (probabilities recomputed): train.hip;
torch owns the memory and does the B-element scalar arithmetic (per-dialog means, the clipping coefficient).
What is trained: everything but the frozen context encoder (model.py:114-115): emb, time_embeds, classifier (or tied), adapter,
the decoder.  ``decoder.embeddings.word_embeddings`` and ``decoder.pooler`` never receive a gradient in the reference (inputs
arrive as ``inputs_embeds``; the pooler output is unused), AdamW skips them, and so does this trainer.
Dropout (configs/train.yaml:22-23: hidden_dropout_prob / attention_probs_dropout_prob of the decoder's BertConfig): applied at
HF's sites -- after the embeddings LayerNorm, on the attention probabilities of self- and cross-attention, after each of the three
output projections before the residual -- with Philox masks regenerated in the backward from (seed, site offset); p = 0 (the
default, and the setting of the gradient fixtures) skips the kernels.  The masks cannot equal torch's generator stream, so with
p > 0 parity is statistical plus exact agreement with a torch emulation fed the same masks (tests/test_gpu_trainer.py).
"""
import math

import torch

from . import _lib, ops
from .model import DiDiSampler


def lr_linear(step, max_lr, min_lr, warmup, decay_range):
    """src/diffusion/model.py:300-308."""
    if min_lr is None:
        raise ValueError("min_lr must be set for the linear scheduler")
    if step < warmup:
        return max_lr
    if step > decay_range:
        return min_lr
    return (max_lr - min_lr) * (decay_range - step) / (decay_range - warmup) + min_lr


def lr_rsqrt_with_warmup(step, max_lr, min_lr, warmup):
    """src/diffusion/model.py:271-297."""
    if warmup != 0 and step < warmup:
        return max_lr * math.sqrt(step / warmup)
    lr = max_lr * math.sqrt(1 / step) if warmup == 0 else max_lr * math.sqrt(warmup / step)
    return max(lr, min_lr) if min_lr is not None else lr


DEFAULT_OPT = dict(weight_decay=0.01, scheduler="linear", lr=1e-4, warmup_steps=2500, min_lr=0.0, decay_range=100_000)  # train.yaml:33-39


class _W:
    """One Linear of the trained graph: fp32 master weight/bias (views into the trainer's parameter dict) + split planes.
    ``flat`` = (flat fp32 parameters, flat hi planes, flat lo planes, {name: offset}) and ``names`` = (weight names, bias names):
    where the stacked tensors (q | k | v) lie back to back in the flat buffer, w, b, hi and lo are VIEWS of the flat buffers -- the
    trainer re-splits all weights with one launch over the flat buffer and nothing is concatenated per step."""

    def __init__(self, weights, biases, pd, flat=None, names=None):
        self.weights, self.biases, self.pd = weights, biases, pd
        self.views = False
        if flat is not None and names is not None:
            fp, fh, fl, offs = flat

            def span(ns, ts):  # (offset, numel) of tensors lying back to back in the flat buffer, else None
                if any(n not in offs for n in ns):
                    return None
                o = [offs[n] for n in ns]
                if any(o[i] + ts[i].numel() != o[i + 1] for i in range(len(ns) - 1)):
                    return None
                return o[0], sum(t_.numel() for t_ in ts)
            sw = span(names[0], weights)
            sb = span(names[1], biases) if biases is not None else (0, 0)
            if sw is not None and sb is not None:
                shape = (sum(w_.shape[0] for w_ in weights), weights[0].shape[1])
                self.w = fp[sw[0]: sw[0] + sw[1]].view(shape)
                self.hi, self.lo = fh[sw[0]: sw[0] + sw[1]].view(shape), fl[sw[0]: sw[0] + sw[1]].view(shape)
                self.b = None if biases is None else fp[sb[0]: sb[0] + sb[1]]
                self.views = True
        if not self.views:
            self.refresh()

    def refresh(self):
        if self.views:  # the trainer's flat split keeps them current
            return
        self.w = self.weights[0] if len(self.weights) == 1 else torch.cat(self.weights, 0)
        self.b = None if self.biases is None else (self.biases[0] if len(self.biases) == 1 else torch.cat(self.biases, 0))
        self.hi, self.lo = ops.split(self.w, dtype=self.pd)


class DiDiTrainer:
    """Holds the fp32 master parameters under the reference's state-dict names and runs training steps on them.

    ``precision``: bf16x3 (fp32-class: what the gradient fixtures are checked in) | fp16x2 | fp16 | bf16 for the decoder GEMMs of
    forward and backward; the frozen encoder always runs bf16x3."""

    def __init__(self, state_dict, cfg, device="cuda", precision="bf16x3", anchor_loss=False, noise_factor=1.0, opt_kwargs=None,
                 gradient_clip_val=1.0, hidden_dropout=0.1, attention_dropout=0.1, freeze_encoder=True):
        """``hidden_dropout`` / ``attention_dropout``: the decoder's ``hidden_dropout_prob`` / ``attention_probs_dropout_prob``;
        the defaults are what the reference trains with (configs/train.yaml:22-23); the gradient fixtures pass 0.0 explicitly.
        ``freeze_encoder``: only True is built (configs/train.yaml:19 and configs/reddit_pretrain.yaml both freeze it)."""
        if not freeze_encoder:
            raise NotImplementedError("freeze_encoder=False (src/diffusion/model.py:112-115,184-188 back-propagates into the context "
                                      "encoder) is not built: the backward stops at the adapter input; both shipped configs freeze the encoder")
        self.cfg, self.device, self.precision = cfg, torch.device(device), precision
        # device fault word of the step's GEMMs: fp16-plane precisions only (bf16 planes cannot overflow where fp32 does not)
        self.fault = ops.new_fault(self.device) if ops.plane_dtype(precision) == torch.float16 else None
        self.hidden_dropout, self.attention_dropout = float(hidden_dropout), float(attention_dropout)
        self.anchor_loss, self.noise_factor = bool(anchor_loss), float(noise_factor)
        self.opt = dict(DEFAULT_OPT, **(opt_kwargs or {}))
        self.clip = gradient_clip_val
        self.frozen = DiDiSampler(state_dict, cfg, device=device, precision="bf16x3", encoder_only=True)  # context encoder, schedule buffers
        self.sigmas, self.std_0, self.T = self.frozen.sigmas, self.frozen.std_0, cfg.diffusion_steps
        self._pd = ops.plane_dtype(precision)
        self._state_extra = {k: v for k, v in state_dict.items() if k.startswith("encoder.") or k in ("sigmas", "std_0")
                             or k.startswith("decoder.pooler.") or k == "decoder.embeddings.word_embeddings.weight"
                             or k == "decoder.embeddings.position_ids"}
        f = lambda k: state_dict[k].detach().to(device=self.device, dtype=torch.float32).contiguous().clone()  # noqa: E731
        P = {}
        names = ["emb.weight", "time_embeds.weight", "decoder.embeddings.position_embeddings.weight",
                 "decoder.embeddings.token_type_embeddings.weight", "decoder.embeddings.LayerNorm.weight", "decoder.embeddings.LayerNorm.bias"]
        if not cfg.tie_weights:
            names.append("classifier.weight")
        if cfg.has_adapter:
            names += ["adapter.0.weight", "adapter.0.bias", "adapter.2.weight", "adapter.2.bias"]
        self._layer_names = []
        for i in range(cfg.dec_layers):
            p = f"decoder.encoder.layer.{i}."
            ln = dict(
                q=p + "attention.self.query", k=p + "attention.self.key", v=p + "attention.self.value", o=p + "attention.output.dense",
                ln1=p + "attention.output.LayerNorm", cq=p + "crossattention.self.query", ck=p + "crossattention.self.key",
                cv=p + "crossattention.self.value", co=p + "crossattention.output.dense", ln2=p + "crossattention.output.LayerNorm",
                up=p + "intermediate.dense", down=p + "output.dense", ln3=p + "output.LayerNorm")
            self._layer_names.append(ln)
            # stacked projections (q | k | v, cross k | v) back to back, weights then biases: their stack is a VIEW of the flat buffer
            for grp in (("q", "k", "v"), ("o",), ("ln1",), ("cq",), ("ck", "cv"), ("co",), ("ln2",), ("up",), ("down",), ("ln3",)):
                names += [ln[g] + ".weight" for g in grp] + [ln[g] + ".bias" for g in grp]
        # one flat fp32 buffer for all trainable parameters (and one each for AdamW's moments): the parameter dict holds views,
        # so the optimiser is ONE launch over the buffer and the gradient all-reduce walks contiguous buckets of its twin
        sizes = [state_dict[n].numel() for n in names]
        pad = lambda n_: (n_ + 3) // 4 * 4  # noqa: E731  (every view 16-byte aligned)
        self._offsets, off = {}, 0
        for n, sz in zip(names, sizes):
            self._offsets[n] = off
            off += pad(sz)
        self._n_flat = off
        self._flat_p = torch.zeros(off, dtype=torch.float32, device=self.device)
        for n in names:
            src = f(n)
            P[n] = self._flat_p[self._offsets[n]: self._offsets[n] + src.numel()].view(src.shape)
            P[n].copy_(src)
        self.params = P
        self._flat_m, self._flat_v = torch.zeros_like(self._flat_p), torch.zeros_like(self._flat_p)
        view = lambda flat: {n: flat[self._offsets[n]: self._offsets[n] + P[n].numel()].view(P[n].shape) for n in names}  # noqa: E731
        self.m, self.v = view(self._flat_m), view(self._flat_v)
        self.step_count = 0
        self._flat_hi = torch.empty(off, dtype=self._pd, device=self.device)  # planes of the whole flat buffer: one split per step
        self._flat_lo = torch.empty_like(self._flat_hi)
        ops.split(self._flat_p, dtype=self._pd, out=(self._flat_hi, self._flat_lo))
        self._build_linears()

    # ------------------------------------------------------------------ parameter views
    def _build_linears(self):
        P, pd = self.params, self._pd
        flat = (self._flat_p, self._flat_hi, self._flat_lo, self._offsets)
        W = lambda *ns: _W([P[n + ".weight"] for n in ns], [P[n + ".bias"] for n in ns], pd, flat,  # noqa: E731
                           ([n + ".weight" for n in ns], [n + ".bias" for n in ns]))
        self._lin = []
        for ln in self._layer_names:
            self._lin.append(dict(qkv=W(ln["q"], ln["k"], ln["v"]), o=W(ln["o"]), cq=W(ln["cq"]), ckv=W(ln["ck"], ln["cv"]), co=W(ln["co"]),
                                  up=W(ln["up"]), down=W(ln["down"])))
        cls_name = "emb.weight" if self.cfg.tie_weights else "classifier.weight"
        self._cls = _W([P[cls_name]], None, pd, flat, ([cls_name], None))
        if self.cfg.has_adapter:
            self._ad0, self._ad2 = W("adapter.0"), W("adapter.2")

    def refresh_planes(self):
        """Re-split every weight after the optimiser moved the fp32 masters: one launch over the flat buffer (the Linears hold views
        of its planes); a Linear whose tensors do not lie back to back re-splits itself."""
        ops.split(self._flat_p, dtype=self._pd, out=(self._flat_hi, self._flat_lo))
        for d in self._lin:
            for w in d.values():
                w.refresh()
        self._cls.refresh()
        if self.cfg.has_adapter:
            self._ad0.refresh()
            self._ad2.refresh()

    def state_dict(self):
        """The reference's checkpoint key layout: trained tensors + the untouched rest."""
        sd = dict(self._state_extra)
        sd.update({k: v.detach().clone() for k, v in self.params.items()})
        if self.cfg.tie_weights:
            sd["classifier.weight"] = sd["emb.weight"]
        return sd

    def sampler(self, **kw):
        """A DiDiSampler on the current weights (validation_step's ``sample``).  Default precision bf16x3 (fp32-class, as the
        reference validates in fp32); pass precision="fp16-sched" for the throughput default of bench.py."""
        return DiDiSampler(self.state_dict(), self.cfg, device=self.device, **kw)

    # ------------------------------------------------------------------ checkpoint / resume (src/training.py:27-33,47)
    CKPT_KEY = "didi_amd_trainer"

    def save_checkpoint(self, path):
        """What Lightning's ``ModelCheckpoint`` writes every ``save_interval`` steps (src/training.py:27-33), as plain tensors
        and numbers only -- ``torch.load(weights_only=True)`` and didi_amd.checkpoint.load_state read it: ``state_dict`` in the
        reference's key layout (so the file loads into ``DiDiSampler`` / ``load_lightning_checkpoint`` like a reference
        checkpoint), ``global_step``, plain ``hyper_parameters``, and under ``didi_amd_trainer`` what a resume needs beyond the
        weights: AdamW's two moments per parameter name, the step count (it drives the LR lambda and the bias correction),
        ``opt_kwargs``, precision, dropout rates, clip value.  Lightning's own ``optimizer_states`` / ``lr_schedulers`` layout
        (param-group indices instead of names) is not written: DESIGN.md section 8."""
        cfg = self.cfg
        torch.save({
            "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
            "global_step": int(self.step_count),
            "hyper_parameters": {"diffusion_steps": cfg.diffusion_steps, "schedule": cfg.schedule, "step_freq": cfg.step_freq,
                                 "sampling_mode": cfg.sampling_mode, "tie_weights": bool(cfg.tie_weights), "anchor_loss": self.anchor_loss,
                                 "noise_factor": self.noise_factor},
            self.CKPT_KEY: {"exp_avg": {k: v.detach().cpu() for k, v in self.m.items()},
                            "exp_avg_sq": {k: v.detach().cpu() for k, v in self.v.items()},
                            "step_count": int(self.step_count), "opt_kwargs": dict(self.opt), "precision": self.precision,
                            "hidden_dropout": self.hidden_dropout, "attention_dropout": self.attention_dropout,
                            "gradient_clip_val": self.clip},
        }, path)

    def load_checkpoint(self, path):
        """Resume from ``save_checkpoint``'s file (``trainer.fit(ckpt_path=resume)``, src/training.py:47): weights, both AdamW
        moments and the step count are restored in place, so the next ``training_step`` is the one an uninterrupted run would
        have taken.  The optimiser settings in the file must be the trainer's (a resumed run that silently changed its LR
        schedule would not be a resume)."""
        ck = torch.load(path, map_location="cpu", weights_only=True)
        st = ck.get(self.CKPT_KEY)
        if st is None:
            raise ValueError(f"{path}: no '{self.CKPT_KEY}' section (a weights-only checkpoint cannot resume an optimiser)")
        if dict(st["opt_kwargs"]) != dict(self.opt) or st["precision"] != self.precision:
            raise ValueError(f"{path}: saved with opt_kwargs {st['opt_kwargs']} / precision {st['precision']}, this trainer has {self.opt} / {self.precision}")
        sd = ck["state_dict"]
        missing = [k for k in self.params if k not in sd or k not in st["exp_avg"] or k not in st["exp_avg_sq"]]
        if missing:
            raise ValueError(f"{path}: missing {missing[:4]}...")
        for k, p_ in self.params.items():
            if tuple(sd[k].shape) != tuple(p_.shape):
                raise ValueError(f"{path}: {k} has shape {tuple(sd[k].shape)}, expected {tuple(p_.shape)}")
            p_.copy_(sd[k])
            self.m[k].copy_(st["exp_avg"][k])
            self.v[k].copy_(st["exp_avg_sq"][k])
        self.step_count = int(st["step_count"])
        self.refresh_planes()
        return self

    # ------------------------------------------------------------------ forward + backward
    def _linear(self, w, a_hi, a_lo, **kw):
        return ops.gemm(a_hi, a_lo, w.hi, w.lo, bias=w.b, precision=self.precision, out_f32=True, fault=self.fault, **kw)

    def _linear_bwd(self, dC, A, w, residual=None, need_dA=True, A_planes=None, dC_drop=None, dC_gelu_u=None):
        return ops.linear_backward(dC, A, w.w, precision=self.precision, residual=residual, need_dA=need_dA, fault=self.fault, A_planes=A_planes,
                                   W_planes=(w.hi, w.lo), dC_drop=dC_drop, dC_gelu_u=dC_gelu_u)

    def check_fault(self, what="training step"):
        """fp16 planes only: wait for the stream and raise if a GEMM of the step wrote a value beyond fp16's range into a plane
        (the planes saturate at +-65504: under the fixed loss scale a saturated activation or gradient would otherwise pass
        unnoticed).  forward_backward calls it once per step."""
        if self.fault is not None:
            try:
                _lib.check_fault(self.fault, what)
            finally:
                self.fault.zero_()

    @torch.no_grad()
    def forward_backward(self, raw_context, target, *, draws=None, t=None, keep=False, dropout_seed=None, grad_sink=None):
        """One ``training_step`` and its backward -> dict(mse, ce, t_T, loss, ce_hat [0-dim device tensors], grads {name: fp32}).

        ``draws`` / ``t``: injected normal draws [n0, n1] and timesteps (parity tests), else Philox noise / ``torch.randint`` as
        the reference draws them.  ``keep``: also return the tensors the tests compare (x_0, x_t, x_0_hat, d x_0_hat).
        ``dropout_seed``: key of the step's dropout masks (default: drawn from torch's global generator).
        ``grad_sink``: an ``_OverlappedReducer``: told after every decoder layer's backward (and at the end) which gradients exist,
        so that their all-reduce runs under the rest of the backward; the gradients it receives still carry the loss scale."""
        cfg, dev, P = self.cfg, self.device, self.params
        ids, tmask = target.input_ids.to(dev), target.attention_mask.to(dev)
        cids, cmask = raw_context.input_ids.to(dev), raw_context.attention_mask.to(dev)
        B, S = ids.shape
        Cn = cids.shape[1]
        H, T, A = cfg.dec_dim, self.T, cfg.dec_heads
        M = B * S
        eps = cfg.dec_ln_eps
        pd = self._pd
        G = {}
        ph, pa = self.hidden_dropout, self.attention_dropout
        dseed = 0
        if ph > 0 or pa > 0:
            dseed = int(dropout_seed) if dropout_seed is not None else int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        n_sites = [0]

        def site():  # a disjoint Philox counter range per dropout site of the step (2^34 counters = 2^36 elements each)
            n_sites[0] += 1
            return n_sites[0] << 34

        def drop(x, residual, off):  # hidden dropout (+ residual), or its backward with residual = None
            return ops.dropout_add(x, residual, ph, dseed, off)
        # ---- noising (model.py:202-207)
        emb = ops.embed_gather(ids, P["emb.weight"])
        seed = 0
        if draws is None:
            seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        if t is None:
            t = torch.randint(1, T + 1, size=(B, 1), device=dev)
        t = t.to(dev)
        # the two host scalars of the step (loss weights are kernel arguments): read HERE, where they depend on the inputs only and the
        # stream is still empty, instead of after the forward, where the read drained a queue of ~700 launches
        n_tok_f = float(tmask.sum().item())
        pf = float((t.view(-1) == 1).float().mean().item())
        sigma = self.sigmas[t.view(-1)]
        s_b = (sigma * self.noise_factor ** 2).contiguous()
        c_b = (1.0 / torch.sqrt(sigma * sigma + 1.0)).contiguous()
        n0 = n1 = None
        if draws is not None:
            n0, n1 = draws[0].to(dev).float().contiguous(), draws[1].to(dev).float().contiguous()
        x_0, x_t = ops.train_noise(emb, s_b, c_b, self.std_0, n0, n1, seed=seed)
        # ---- context: frozen encoder, trainable adapter (model.py:165-170, 183-188)
        c_hi, c_lo, enc = self.frozen._encode_raw(cids, cmask)
        mask32 = cmask.to(torch.int32).contiguous()
        if cfg.has_adapter:
            if pd != torch.bfloat16:
                c_hi, c_lo = ops.split(enc, dtype=pd)
            z_hi, z_lo, z = self._linear(self._ad0, c_hi, c_lo, act="tanh")
            x_hi, x_lo, ctx = self._linear(self._ad2, z_hi, z_lo)
        else:
            ctx = enc
            x_hi, x_lo = (c_hi, c_lo) if pd == torch.bfloat16 else ops.split(enc, dtype=pd)
        # ---- decoder forward, activations kept in fp32 for the backward
        x_in = ops.train_embed_sum(x_t, P["time_embeds.weight"], t, P["decoder.embeddings.position_embeddings.weight"],
                                   P["decoder.embeddings.token_type_embeddings.weight"][0].contiguous())
        h_hi, h_lo, h = ops.layernorm(x_in.view(M, H), P["decoder.embeddings.LayerNorm.weight"], P["decoder.embeddings.LayerNorm.bias"], eps,
                                      f32=True, dtype=pd)
        off_emb = site()
        if ph > 0:  # BertEmbeddings.dropout (HF:bert:107)
            h = drop(h, None, off_emb)
            h_hi, h_lo = ops.split(h, dtype=pd)
        saved = []
        x3 = self.precision in ("bf16x3", "fp16x3")
        for li, (L, ln) in enumerate(zip(self._lin, self._layer_names)):
            s = dict(h0=h, h0p=(h_hi, h_lo))  # *p: the planes the forward multiplied with -- the weight gradient's operand as it is
            _, _, qkv = self._linear(L["qkv"], h_hi, h_lo, out_planes=False)
            q3 = qkv.view(B, S, 3 * H)
            s["off_sa"] = site()
            att = ops.attention_train(q3[:, :, :H], q3[:, :, H:2 * H], q3[:, :, 2 * H:], A, causal=True, drop_p=pa, seed=dseed,
                                      offset=s["off_sa"]).view(M, H)
            a_hi, a_lo = ops.split(att, dtype=pd)
            s["qkv"], s["att"], s["attp"] = qkv, att, (a_hi, a_lo)
            s["off_o"] = site()
            if ph > 0:  # BertSelfOutput: LayerNorm(dropout(dense(x)) + input)
                _, _, y = self._linear(L["o"], a_hi, a_lo, out_planes=False)
                h_hi, h_lo, h1, s1 = ops.dropout_add_layernorm(y, h, ph, dseed, s["off_o"], P[ln["ln1"] + ".weight"], P[ln["ln1"] + ".bias"], eps, dtype=pd)
            else:
                _, _, s1 = self._linear(L["o"], a_hi, a_lo, residual=h, out_planes=False)
                h_hi, h_lo, h1 = ops.layernorm(s1, P[ln["ln1"] + ".weight"], P[ln["ln1"] + ".bias"], eps, f32=True, dtype=pd)
            s["s1"], s["h1"], s["h1p"] = s1, h1, (h_hi, h_lo)
            _, _, cq = self._linear(L["cq"], h_hi, h_lo, out_planes=False)
            _, _, ckv = self._linear(L["ckv"], x_hi, x_lo, out_planes=False)
            kv3 = ckv.view(B, Cn, 2 * H)
            s["off_ca"] = site()
            catt = ops.attention_train(cq.view(B, S, H), kv3[:, :, :H], kv3[:, :, H:], A, causal=False, key_mask=mask32, drop_p=pa,
                                       seed=dseed, offset=s["off_ca"]).view(M, H)
            ca_hi, ca_lo = ops.split(catt, dtype=pd)
            s["cq"], s["ckv"], s["catt"], s["cattp"] = cq, ckv, catt, (ca_hi, ca_lo)
            s["off_co"] = site()
            if ph > 0:
                _, _, y = self._linear(L["co"], ca_hi, ca_lo, out_planes=False)
                h_hi, h_lo, h2, s2 = ops.dropout_add_layernorm(y, h1, ph, dseed, s["off_co"], P[ln["ln2"] + ".weight"], P[ln["ln2"] + ".bias"], eps, dtype=pd)
            else:
                _, _, s2 = self._linear(L["co"], ca_hi, ca_lo, residual=h1, out_planes=False)
                h_hi, h_lo, h2 = ops.layernorm(s2, P[ln["ln2"] + ".weight"], P[ln["ln2"] + ".bias"], eps, f32=True, dtype=pd)
            s["s2"], s["h2"], s["h2p"] = s2, h2, (h_hi, h_lo)
            _, _, u = self._linear(L["up"], h_hi, h_lo, out_planes=False)
            f_hi, f_lo, f_ = self._gelu_planes(u, H)
            s["u"], s["f"], s["fp"] = u, f_, (f_hi, f_lo)
            s["off_down"] = site()
            if ph > 0:
                _, _, y = self._linear(L["down"], f_hi, f_lo, out_planes=False)
                h_hi, h_lo, h, s3 = ops.dropout_add_layernorm(y, h2, ph, dseed, s["off_down"], P[ln["ln3"] + ".weight"], P[ln["ln3"] + ".bias"], eps, dtype=pd)
            else:
                _, _, s3 = self._linear(L["down"], f_hi, f_lo, residual=h2, out_planes=False)
                h_hi, h_lo, h = ops.layernorm(s3, P[ln["ln3"] + ".weight"], P[ln["ln3"] + ".bias"], eps, f32=True, dtype=pd)
            s["s3"] = s3
            saved.append(s)
        x_0_hat = h.view(B, S, H)
        # ---- losses (model.py:216-243)
        sums = ops.train_loss_terms(x_0_hat, emb, x_0, tmask)
        p = (t.view(-1) == 1).float().mean()
        per = sums / float(S * H)
        mse = p * per[:, 0].mean() + (1.0 - p) * per[:, 1].mean()
        sigma_T = self.sigmas[-1]
        t_T = sums[:, 2].sum() / (sigma_T * sigma_T + 1.0) / float(B * S * H)
        # Loss scale of the backward: every loss weight carries a 1 / (B S H) or 1 / tokens, which puts the gradients of the
        # activations near fp16's subnormals.  fp16-plane precisions run the backward on `ls` x the loss (a power of two, so
        # exact) and the parameter gradients are scaled back at the end; bf16 planes have fp32's exponent range (ls = 1).
        ls = 2.0 ** round(math.log2(B * S * H)) if pd == torch.float16 else 1.0

        def logits_of(x2d):
            hi, lo = ops.split(x2d, dtype=pd)
            _, _, lg = ops.gemm(hi, lo, self._cls.hi, self._cls.lo, precision=self.precision, out_f32=True, out_planes=False, fault=self.fault)
            return lg

        def ce_and_grad(x2d):
            """cross-entropy of classifier(x) and, in place of the logits, its gradient with respect to them."""
            lg = logits_of(x2d)
            _, nll = ops.ce_backward(lg, ids, tmask, ls / n_tok_f)
            return nll.sum() / n_tok_f, lg

        hat2d = x_0_hat.reshape(M, H)
        if self.anchor_loss:
            ce, dlg = ce_and_grad(hat2d)
            ce_hat = ce
            d_hat_ce, dWc, _ = ops.linear_backward(dlg, hat2d, self._cls.w, precision=self.precision, bias=False)
            extra = None
        else:
            ce, dlg = ce_and_grad(x_0.reshape(M, H))
            extra, dWc, _ = ops.linear_backward(dlg, x_0.reshape(M, H), self._cls.w, precision=self.precision, bias=False)
            hi, lo = ops.split(hat2d, dtype=pd)
            _, nll = ops.round_ce(hi, lo if x3 else None, self._cls.hi, self._cls.lo, ids, self.precision)
            m = tmask.reshape(-1).to(nll.dtype)
            ce_hat = (nll * m).sum() / m.sum()
            d_hat_ce = None
        del dlg
        loss = mse + ce if self.anchor_loss else mse + ce + t_T
        # ---- backward
        w_emb, w_x0 = ls * 2.0 * pf / (B * S * H), ls * 2.0 * (1.0 - pf) / (B * S * H)
        w_T = 0.0 if self.anchor_loss else ls * 2.0 / ((float(sigma_T) ** 2 + 1.0) * B * S * H)
        dh = ops.train_mse_backward(x_0_hat, emb, x_0, tmask, w_emb, w_x0).view(M, H)
        if d_hat_ce is not None:
            ops.axpby(1.0, d_hat_ce, 1.0, dh)
        d_x0_hat = dh.clone() if keep else None
        dctx = None  # gradient of the context (adapter output): summed over the layers' cross-attention K/V projections
        need_dctx = cfg.has_adapter
        for li in reversed(range(cfg.dec_layers)):
            L, ln, s = self._lin[li], self._layer_names[li], saved[li]

            def put(name, dw, db):
                G[name + ".weight"], G[name + ".bias"] = dw, db

            ds3, dg, db = ops.layernorm_backward(dh, s["s3"], P[ln["ln3"] + ".weight"], eps)
            put(ln["ln3"], dg, db)
            df, dW, dbias = self._linear_bwd(ds3, s["f"], L["down"], A_planes=s["fp"], dC_drop=(ph, dseed, s["off_down"]))
            put(ln["down"], dW, dbias)
            # GELU's backward rides on the pass that splits the gradient; + the skip connection around the FFN
            dh2, dW, dbias = self._linear_bwd(df, s["h2"], L["up"], residual=ds3, A_planes=s["h2p"], dC_gelu_u=s["u"])
            put(ln["up"], dW, dbias)
            ds2, dg, db = ops.layernorm_backward(dh2, s["s2"], P[ln["ln2"] + ".weight"], eps)
            put(ln["ln2"], dg, db)
            dcatt, dW, dbias = self._linear_bwd(ds2, s["catt"], L["co"], A_planes=s["cattp"], dC_drop=(ph, dseed, s["off_co"]))
            put(ln["co"], dW, dbias)
            ckv = s["ckv"].view(B, Cn, 2 * H)
            dckv = torch.empty_like(ckv)
            dcq = torch.empty((B, S, H), dtype=torch.float32, device=dev)
            ops.attention_backward(s["cq"].view(B, S, H), ckv[:, :, :H], ckv[:, :, H:], dcatt.view(B, S, H), A, causal=False, key_mask=mask32,
                                   out=(dcq, dckv[:, :, :H], dckv[:, :, H:]), drop_p=pa, seed=dseed, offset=s["off_ca"])
            dh1, dW, dbias = self._linear_bwd(dcq.view(M, H), s["h1"], L["cq"], residual=ds2, A_planes=s["h1p"])
            put(ln["cq"], dW, dbias)
            dctx_l, dW, dbias = self._linear_bwd(dckv.view(B * Cn, 2 * H), ctx.view(B * Cn, H), L["ckv"], residual=dctx, need_dA=need_dctx, A_planes=(x_hi, x_lo))
            put(ln["ck"], dW[:H].contiguous(), dbias[:H].contiguous())
            put(ln["cv"], dW[H:].contiguous(), dbias[H:].contiguous())
            if need_dctx:
                dctx = dctx_l  # accumulated over the layers in the dgrad epilogue
            ds1, dg, db = ops.layernorm_backward(dh1, s["s1"], P[ln["ln1"] + ".weight"], eps)
            put(ln["ln1"], dg, db)
            datt, dW, dbias = self._linear_bwd(ds1, s["att"], L["o"], A_planes=s["attp"], dC_drop=(ph, dseed, s["off_o"]))
            put(ln["o"], dW, dbias)
            qkv = s["qkv"].view(B, S, 3 * H)
            dqkv = torch.empty_like(qkv)
            ops.attention_backward(qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:], datt.view(B, S, H), A, causal=True,
                                   out=(dqkv[:, :, :H], dqkv[:, :, H:2 * H], dqkv[:, :, 2 * H:]), drop_p=pa, seed=dseed, offset=s["off_sa"])
            dh0, dW, dbias = self._linear_bwd(dqkv.view(M, 3 * H), s["h0"], L["qkv"], residual=ds1, A_planes=s["h0p"])
            for j, key in enumerate(("q", "k", "v")):
                put(ln[key], dW[j * H:(j + 1) * H].contiguous(), dbias[j * H:(j + 1) * H].contiguous())
            dh = dh0
            saved[li] = None  # free this layer's activations
            if grad_sink is not None:
                grad_sink.layer_done(li, G)
        # decoder embeddings
        if ph > 0:
            dh = drop(dh, None, off_emb)
        dx_in, dg, db = ops.layernorm_backward(dh, x_in.view(M, H), P["decoder.embeddings.LayerNorm.weight"], eps)
        G["decoder.embeddings.LayerNorm.weight"], G["decoder.embeddings.LayerNorm.bias"] = dg, db
        gpos = torch.zeros_like(P["decoder.embeddings.position_embeddings.weight"])
        gpos[:S] = ops.colsum(dx_in.view(B, S * H)).view(S, H)
        G["decoder.embeddings.position_embeddings.weight"] = gpos
        gtok = torch.zeros_like(P["decoder.embeddings.token_type_embeddings.weight"])
        gtok[0] = ops.colsum(dx_in.view(M, H))
        G["decoder.embeddings.token_type_embeddings.weight"] = gtok
        gte = torch.zeros_like(P["time_embeds.weight"])
        ops.embedding_backward(t.view(-1), ops.seq_sum(dx_in.view(B, S, H)), gte)
        G["time_embeds.weight"] = gte
        # target embeddings: mse (both branches), t_T, x_t, cross-entropy on classifier(x_0)
        d_emb = ops.train_target_backward(x_0_hat, emb, x_0, tmask, w_emb, w_x0, w_T, c_b, dx_in.view(B, S, H),
                                          None if extra is None else extra.view(B, S, H))
        gemb = torch.zeros_like(P["emb.weight"])
        ops.embedding_backward(ids, d_emb, gemb, skip_id=cfg.pad_idx)
        if cfg.tie_weights:
            ops.axpby(1.0, dWc, 1.0, gemb)
        else:
            G["classifier.weight"] = dWc
        G["emb.weight"] = gemb
        # adapter (model.py:122-123)
        if cfg.has_adapter:
            dz, dW, dbias = self._linear_bwd(dctx, z, self._ad2)
            G["adapter.2.weight"], G["adapter.2.bias"] = dW, dbias
            dpre = ops.act_backward("tanh", dz, z)
            _, dW, dbias = ops.linear_backward(dpre, enc, self._ad0.w, precision=self.precision, need_dA=False)
            G["adapter.0.weight"], G["adapter.0.bias"] = dW, dbias
        if grad_sink is not None:
            grad_sink.finish(G, ls)
        elif ls != 1.0:
            for g_ in G.values():
                ops.axpby(0.0, g_, 1.0 / ls, g_)
        self.check_fault()
        out = dict(mse=mse, ce=ce, t_T=t_T, loss=loss, ce_hat=ce_hat, grads=G, t=t)
        if keep:
            out.update(emb=emb, x_0=x_0, x_t=x_t, x_0_hat=x_0_hat, d_x_0_hat=d_x0_hat.view(B, S, H) / ls)
        return out

    def _gelu_planes(self, u, H):
        """gelu(u) as planes (+ fp32 only where the weight gradient of the next Linear still wants its input in fp32): the exact erf
        form of the GEMM epilogue, here as its own pass because the backward keeps the PRE-activation."""
        M, F = u.shape
        return ops.act_forward("gelu", u, dtype=self._pd, f32=not ops.wgrad_supported(M, H, F, self.precision))

    # ------------------------------------------------------------------ optimiser (model.py:136-158, train.yaml:33-45)
    def lr_at(self, step):
        o = self.opt
        if o["scheduler"] == "linear":
            return lr_linear(step, o["lr"], o["min_lr"], o["warmup_steps"], o["decay_range"])
        return lr_rsqrt_with_warmup(step, o["lr"], o["min_lr"], o["warmup_steps"])

    METRIC_SLOTS = 8  # tail of the flat gradient buffer: the five logged scalars ride in the last all-reduce bucket

    @torch.no_grad()
    def flatten_gradients(self, grads, metrics=None):
        """{name: fp32} -> one flat buffer laid out like the parameters (+ METRIC_SLOTS tail floats holding ``metrics``)."""
        flat = torch.zeros(self._n_flat + self.METRIC_SLOTS, dtype=torch.float32, device=self.device)
        dst, src = [], []
        for n, off in self._offsets.items():
            g_ = grads[n]
            assert g_.shape == self.params[n].shape, n
            dst.append(flat[off: off + g_.numel()].view(g_.shape))
            src.append(g_)
        torch._foreach_copy_(dst, src)  # memory plumbing: a handful of batched copy launches instead of one per tensor
        if metrics is not None:
            flat[self._n_flat: self._n_flat + metrics.numel()] = metrics
        return flat

    def allreduce_flat(self, flat, world, bucket_bytes=64 << 20):
        """Mean over ranks of the flat gradient buffer (metric tail included), in place: didi_amd.dist.allreduce_flat."""
        from . import dist as ddist

        return ddist.allreduce_flat(flat, world, bucket_bytes=bucket_bytes, tail=self.METRIC_SLOTS)

    @torch.no_grad()
    def optimizer_step(self, grads):
        """Clip the global gradient norm to ``gradient_clip_val`` (Lightning's clip_grad_norm_: coefficient
        min(1, clip / (norm + 1e-6))), then AdamW(lr = 1 x lambda(step), betas (0.9, 0.999), eps 1e-8) on every parameter.
        LambdaLR semantics: the k-th optimizer.step() (k = 1, 2, ...) runs with lambda(k - 1).
        ``grads``: {name: fp32} or the flat buffer of ``flatten_gradients``.  -> gradient norm (device)."""
        flat = grads if torch.is_tensor(grads) else self.flatten_gradients(grads)
        g = flat[: self._n_flat]
        lr = self.lr_at(self.step_count)
        self.step_count += 1
        total = torch.zeros(1, dtype=torch.float32, device=self.device)
        ops.sumsq_accumulate(g, total)
        norm = total.sqrt()
        gscale = None
        if self.clip is not None:
            gscale = torch.clamp(float(self.clip) / (norm + 1e-6), max=1.0)
        ops.adamw_step(self._flat_p, g, self._flat_m, self._flat_v, lr, 0.9, 0.999, 1e-8, self.opt["weight_decay"], self.step_count, gscale)
        self.refresh_planes()
        return norm

    def training_step(self, raw_context, target, world=1, overlap=True, **kw):
        """forward + backward + (``world`` > 1: bucketed gradient all-reduce over the default process group with the five logged
        scalars in the flat buffer's tail, src/training.py:37 + model.py:241-242) + clip + AdamW -> the step's dict with
        ``metrics`` (fp32 [5]: mse, ce, t_T, loss, ce_hat, averaged over ranks) and ``grad_norm``.
        ``overlap``: launch each bucket's all-reduce as soon as its layers' backward is done (what DDP does), on RCCL's own
        stream under the remaining backward; False = one pass over the buckets after the backward."""
        from .training import metric_message

        if world > 1 and overlap:
            sink = _OverlappedReducer(self, world)
            out = self.forward_backward(raw_context, target, grad_sink=sink, **kw)
            flat = sink.flat
            flat[self._n_flat: self._n_flat + 5] = metric_message(out) / world  # the tail's own small all-reduce (5 floats)
            sink.reduce_tail()
        else:
            out = self.forward_backward(raw_context, target, **kw)
            flat = self.flatten_gradients(out["grads"], metric_message(out))
            self.allreduce_flat(flat, world)
        out["grad_norm"] = self.optimizer_step(flat)
        out["metrics"] = flat[self._n_flat: self._n_flat + 5].clone()
        return out


def layer_buckets(layer_ranges, n_flat, layers_per_bucket=2):
    """Bucket plan of the overlapped reduction.  ``layer_ranges``: [(start, end)] of each decoder layer's slice of the flat buffer
    (ascending, contiguous); the backward finishes layers from the LAST to the first.  -> (buckets, head): ``buckets[j]`` =
    (first_layer, start, end) is launched once layer ``first_layer`` is done and covers ``layers_per_bucket`` layers (the group
    containing layer 0 may be smaller); ``head`` = (0, start_of_layer_0) holds everything that precedes the layers (embeddings,
    classifier, adapter), reduced last.  The slices tile [0, n_flat) exactly."""
    L = len(layer_ranges)
    assert all(a[1] == b[0] for a, b in zip(layer_ranges, layer_ranges[1:])) and layer_ranges[-1][1] == n_flat
    buckets = []
    hi = L
    while hi > 0:
        lo = max(0, hi - layers_per_bucket)
        buckets.append((lo, layer_ranges[lo][0], layer_ranges[hi - 1][1]))
        hi = lo
    return buckets, (0, layer_ranges[0][0])


class _OverlappedReducer:
    """Gradient all-reduce under the backward (what DDP's bucketed reducer does, src/training.py:37).  The flat gradient buffer
    exists from the start of the backward; when the last layer of a bucket (two decoder layers = 57 MB at BERT-base width: xGMI
    ring steps are per-link bound, so few large messages) has its gradients, they are copied into their slice and ONE
    ``all_reduce(async_op=True)`` is enqueued: RCCL runs it on its own stream after the copies, the backward of the earlier
    layers goes on.  The head of the buffer (embeddings, classifier, adapter: produced last) follows at the end, and the
    optimiser waits on every handle.  Averaging (and the fp16 modes' loss scale) is one scaling of the flat buffer afterwards."""

    def __init__(self, trainer, world, all_reduce=None):
        import torch.distributed as tdist

        self.tr, self.world = trainer, world
        self.flat = torch.zeros(trainer._n_flat + trainer.METRIC_SLOTS, dtype=torch.float32, device=trainer.device)
        self._all_reduce = all_reduce or (lambda t_: tdist.all_reduce(t_, op=tdist.ReduceOp.SUM, async_op=True))
        ranges = []
        for ln in trainer._layer_names:
            offs = [trainer._offsets[v + sfx] for v in ln.values() for sfx in (".weight", ".bias")]
            ends = [trainer._offsets[v + sfx] + (trainer.params[v + sfx].numel() + 3) // 4 * 4 for v in ln.values() for sfx in (".weight", ".bias")]
            ranges.append((min(offs), max(ends)))
        self.buckets, self.head = layer_buckets(ranges, trainer._n_flat)
        self._launch_at = {b[0]: b for b in self.buckets}
        self._layer_keys = [[v + sfx for v in ln.values() for sfx in (".weight", ".bias")] for ln in trainer._layer_names]
        self._copied = set()
        self.works = []

    def _copy(self, keys, G):
        dst, src = [], []
        for k in keys:
            off = self.tr._offsets[k]
            dst.append(self.flat[off: off + G[k].numel()].view(G[k].shape))
            src.append(G[k])
        torch._foreach_copy_(dst, src)
        self._copied.update(keys)

    def layer_done(self, li, G):
        self._copy(self._layer_keys[li], G)
        b = self._launch_at.get(li)
        if b is not None:
            self.works.append(self._all_reduce(self.flat[b[1]: b[2]]))

    def finish(self, G, loss_scale):
        rest = [k for k in self.tr._offsets if k not in self._copied]
        self._copy(rest, G)
        if self.head[1] > self.head[0]:
            self.works.append(self._all_reduce(self.flat[self.head[0]: self.head[1]]))
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []
        n = self.tr._n_flat
        ops.axpby(0.0, self.flat[:n], 1.0 / (self.world * loss_scale), self.flat[:n])
        for k in G:  # the dict the caller sees: views of the reduced buffer
            off = self.tr._offsets[k]
            G[k] = self.flat[off: off + G[k].numel()].view(G[k].shape)

    def reduce_tail(self):
        w = self._all_reduce(self.flat[self.tr._n_flat:])
        if w is not None:
            w.wait()
