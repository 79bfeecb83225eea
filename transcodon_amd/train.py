"""Training step on the MI355X kernels: autograd bridge, objective, optimiser, schedule.

Mirrors ``CodonModel`` of the reference's pretraining.py:38-160 without Lightning:
``ProteinBertModel.forward`` under grad mode goes through :class:`EncoderTrainFunction`
(tc_encoder_forward_train / tc_encoder_backward), :func:`mlm_loss` is the objective of
``training_step`` (:109-130), :class:`AdamW` + :func:`warmup_cosine` are
``configure_optimizers`` (:64-92).  torch only owns memory, the autograd tape and the
process group.
"""

import ctypes as C
import math

import torch

from . import _lib
from ._lib import EncoderGrads, LayerGrads, check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _align(n, a=64):
    return (n + a - 1) // a * a


def train_parameters(model):
    """The parameters the forward uses, in the order EncoderTrainFunction returns gradients."""
    ps = [model.embed_tokens.weight, model.species_embedding.weight,
          model.emb_layer_norm_after.weight, model.emb_layer_norm_after.bias,
          model.lm_head.dense.weight, model.lm_head.dense.bias,
          model.lm_head.layer_norm.weight, model.lm_head.layer_norm.bias, model.lm_head.bias,
          model.structure_linear.dense.weight, model.structure_linear.dense.bias,
          model.structure_linear.layer_norm.weight, model.structure_linear.layer_norm.bias,
          model.structure_linear.out_proj.weight, model.structure_linear.out_proj.bias,
          model.structure_linear.bias]
    for l in model.layers:
        a = l.self_attn
        ps += [l.self_attn_layer_norm.weight, l.self_attn_layer_norm.bias,
               a.q_proj.weight, a.k_proj.weight, a.v_proj.weight,
               a.q_proj.bias, a.k_proj.bias, a.v_proj.bias,
               a.out_proj.weight, a.out_proj.bias,
               l.final_layer_norm.weight, l.final_layer_norm.bias,
               l.fc1.weight, l.fc1.bias, l.fc2.weight, l.fc2.bias]
    return ps


class PackedLayout:
    """Offsets (in f32 elements) of every trained tensor in the kernels' packed order: q/k/v
    weights stacked as [3E,E], the two head dense layers as [2E,E].  ``st_bias`` is the single
    gradient slot shared by structure_linear.out_proj.bias and structure_linear.bias;
    ``st_bias2`` gives the second of them a slot of its own in a parameter store."""

    HEAD = ("tok_table", "species_table", "lnf_g", "lnf_b", "w_heads", "b_heads", "lm_ln_g", "lm_ln_b",
            "lm_bias", "st_ln_g", "st_ln_b", "st_w", "st_bias", "st_bias2")

    def __init__(self, model):
        E, F = model.args.embed_dim, model.args.ffn_embed_dim
        V, S, L = model.alphabet_size, model.args.species_vocab_size, len(model.layers)
        self.dims = (E, F, V, S, L)
        head = dict(tok_table=V * E, species_table=S * E, lnf_g=E, lnf_b=E, w_heads=2 * E * E, b_heads=2 * E,
                    lm_ln_g=E, lm_ln_b=E, lm_bias=V, st_ln_g=E, st_ln_b=E, st_w=3 * E, st_bias=3, st_bias2=3)
        layer = [("ln1_g", E), ("ln1_b", E), ("w_qkv", 3 * E * E), ("b_qkv", 3 * E), ("w_o", E * E), ("b_o", E),
                 ("ln2_g", E), ("ln2_b", E), ("w_fc1", F * E), ("b_fc1", F), ("w_fc2", E * F), ("b_fc2", E)]
        off, self.hoff, self.loff = 0, {}, []
        for n in self.HEAD:
            self.hoff[n] = (off, head[n])
            off += _align(head[n])
        for _ in range(L):
            d = {}
            for n, sz in layer:
                d[n] = (off, sz)
                off += _align(sz)
            self.loff.append(d)
        self.total = off

    def grads_struct(self, flat):
        """tc_encoder_grads pointing into ``flat``; returns (struct, keep-alive layer array)."""
        base = flat.data_ptr()
        layers = (LayerGrads * len(self.loff))()
        for i, d in enumerate(self.loff):
            for n, (o, _) in d.items():
                setattr(layers[i], n, base + 4 * o)
        g = EncoderGrads()
        for n, (o, _) in self.hoff.items():
            if n != "st_bias2":
                setattr(g, n, base + 4 * o)
        g.layers = C.cast(layers, C.POINTER(LayerGrads))
        return g, layers

    def views(self, flat, own_second_bias):
        """Per-parameter views of ``flat`` in train_parameters order.  own_second_bias: give
        structure_linear.bias the st_bias2 slot (parameter stores) instead of aliasing st_bias."""
        E, F, V, S, _ = self.dims
        hoff = self.hoff
        v = lambda o_sz, *shape: flat[o_sz[0]:o_sz[0] + o_sz[1]].view(*shape)
        w_heads, b_heads = v(hoff["w_heads"], 2 * E, E), v(hoff["b_heads"], 2 * E)
        views = [v(hoff["tok_table"], V, E), v(hoff["species_table"], S, E), v(hoff["lnf_g"], E), v(hoff["lnf_b"], E),
                 w_heads[:E], b_heads[:E], v(hoff["lm_ln_g"], E), v(hoff["lm_ln_b"], E), v(hoff["lm_bias"], V),
                 w_heads[E:], b_heads[E:], v(hoff["st_ln_g"], E), v(hoff["st_ln_b"], E), v(hoff["st_w"], 3, E),
                 v(hoff["st_bias"], 3), v(hoff["st_bias2" if own_second_bias else "st_bias"], 3)]
        for d in self.loff:
            w_qkv, b_qkv = v(d["w_qkv"], 3 * E, E), v(d["b_qkv"], 3 * E)
            views += [v(d["ln1_g"], E), v(d["ln1_b"], E), w_qkv[:E], w_qkv[E:2 * E], w_qkv[2 * E:],
                      b_qkv[:E], b_qkv[E:2 * E], b_qkv[2 * E:], v(d["w_o"], E, E), v(d["b_o"], E),
                      v(d["ln2_g"], E), v(d["ln2_b"], E), v(d["w_fc1"], F, E), v(d["b_fc1"], F),
                      v(d["w_fc2"], E, F), v(d["b_fc2"], E)]
        return views


class _GradPack:
    """One zeroed f32 buffer holding every gradient in the kernels' packed layout, the C
    struct that points into it, and the per-parameter views in train_parameters order."""

    def __init__(self, model, device):
        lay = PackedLayout(model)
        self.flat = torch.zeros(lay.total, dtype=torch.float32, device=device)
        self.struct, self.layers = lay.grads_struct(self.flat)
        self.views = lay.views(self.flat, own_second_bias=False)


class EncoderTrainFunction(torch.autograd.Function):
    """(tokens, species, parameters) -> (logits [B,T,V], struct logits [B,T,3]) with the HIP
    forward; backward fills every parameter gradient with one tc_encoder_backward call."""

    @staticmethod
    def forward(ctx, model, tok32, sp32, *params):
        B, T = tok32.shape
        dev = tok32.device
        dt = model._dtype()
        pk = model._packed(dt, T, dev)
        lib = _lib.load()
        need = lib.tc_train_workspace_bytes(C.byref(pk["desc"]), B, T)
        if need == 0:
            raise _lib.HipLibraryError("tc_train_workspace_bytes: unsupported configuration")
        ws = model._get_workspace(need, dev, train=True)
        logits = torch.empty(B, T, model.alphabet_size, dtype=torch.float32, device=dev)
        struct = torch.empty(B, T, 3, dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.tc_encoder_forward_train(C.byref(pk["desc"]), tok32.data_ptr(), sp32.data_ptr(), B, T,
                                           logits.data_ptr(), struct.data_ptr(), status.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _stream()), "tc_encoder_forward_train")
        if model.check_indices and int(status.item()) != 0:
            raise IndexError("token or species id out of range")
        model._train_generation += 1
        ctx.model, ctx.pk, ctx.ws, ctx.tok32, ctx.sp32 = model, pk, ws, tok32, sp32
        ctx.generation = model._train_generation
        ctx.set_materialize_grads(False)
        return logits, struct

    @staticmethod
    def backward(ctx, dlogits, dstruct):
        model = ctx.model
        if ctx.generation != model._train_generation:
            raise RuntimeError("the saved activations were overwritten by a later forward: run backward "
                               "before the next forward of the same model")
        if dlogits is None and dstruct is None:
            return (None,) * (3 + len(train_parameters(model)))
        B, T = ctx.tok32.shape
        dev = ctx.tok32.device
        if dlogits is None:
            dlogits = torch.zeros(B, T, model.alphabet_size, dtype=torch.float32, device=dev)
        dlogits = dlogits.float().contiguous()
        dstruct = None if dstruct is None else dstruct.float().contiguous()
        store = getattr(model, "_flat_store", None)
        struct = store.grads_struct if store is not None else None
        gp = None
        if struct is None:
            gp = _GradPack(model, dev)
            struct = gp.struct
        sync = getattr(store, "grad_sync", None) if gp is None else None
        args = (C.byref(ctx.pk["desc"]), C.byref(struct), ctx.tok32.data_ptr(), ctx.sp32.data_ptr(), B, T,
                dlogits.data_ptr(), None if dstruct is None else dstruct.data_ptr(), ctx.ws.data_ptr(),
                ctx.ws.numel())
        if sync is not None:
            # per-layer notification: the gradient exchange of a finished layer overlaps the rest
            cb = _lib.GRADS_READY_FN(lambda layer, _user: sync.on_ready(layer))
            try:
                check(_lib.load().tc_encoder_backward_notify(*args, cb, None, _stream()), "tc_encoder_backward")
            except BaseException:
                sync.abandon()      # exchanges already started must not be waited for by the next step
                raise
            sync.finish()
        else:
            check(_lib.load().tc_encoder_backward(*args, _stream()), "tc_encoder_backward")
        params = train_parameters(model)
        if gp is None:
            # the kernels accumulated straight into the parameter store's gradient buffer (which
            # the parameters' .grad attributes view): nothing for autograd to add
            if sync is None:
                store.sync_shared_bias_grad()
            return (None,) * (3 + len(params))
        grads = [g.to(p.dtype) if p.requires_grad else None for p, g in zip(params, gp.views)]
        return (None, None, None, *grads)


class _MlmLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, struct_logits, struct_labels, struct_weight):
        lib = _lib.load()
        M, V = labels.numel(), logits.shape[-1]
        dev = logits.device
        lg = logits.reshape(M, V).float().contiguous()
        lb = labels.reshape(M).to(torch.int32).contiguous()
        use_struct = struct_logits is not None and struct_labels is not None
        S = struct_logits.shape[-1] if use_struct else 0
        sl = struct_logits.reshape(M, S).float().contiguous() if use_struct else None
        sb = struct_labels.reshape(M).to(torch.int32).contiguous() if use_struct else None
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        dl = torch.empty_like(lg)
        ds = torch.empty_like(sl) if use_struct else None
        p = lambda t: None if t is None else t.data_ptr()
        check(lib.tc_mlm_loss(p(lg), p(lb), p(sl), p(sb), p(loss), p(dl), p(ds), p(stats), struct_weight, 1.0,
                              M, V, S, _stream()), "tc_mlm_loss")
        ctx.save_for_backward(dl, ds) if use_struct else ctx.save_for_backward(dl)
        ctx.shapes = (logits.shape, struct_logits.shape if use_struct else None)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        dl = (saved[0] * g).view(ctx.shapes[0])
        ds = (saved[1] * g).view(ctx.shapes[1]) if len(saved) > 1 else None
        return dl, None, ds, None, None


def mlm_loss(logits, labels, struct_logits=None, struct_labels=None, struct_weight=0.4, struct_present=None):
    """pretraining.py:109-130: mean cross-entropy over labelled positions (ignore -100) plus
    ``struct_weight`` times the structure cross-entropy when any structure label is set.
    A label outside [0, V) other than -100 makes the loss NaN (torch raises there).

    ``struct_present``: whether any structure label is set, when the caller knows (a data feed
    without structure labels passes False); None asks the device, which costs a host
    synchronisation per call, like the reference's own ``(struct_label != -100).any()``."""
    if not logits.is_cuda:
        raise _lib.HipLibraryError("mlm_loss runs on the GPU only (no CPU fallback)")
    if struct_labels is None or struct_logits is None:
        struct_present = False
    elif struct_present is None:
        struct_present = bool((struct_labels != -100).any())
    if not struct_present:
        struct_logits = struct_labels = None      # pretraining.py:122: the term exists only with labels
    return _MlmLossFunction.apply(logits, labels, struct_logits, struct_labels, struct_weight)


_CODING_CACHE = {}


def mlm_mask(tokens, lens, seed, struct_labels=None, mask_proportion=0.25, mask_percent=0.8, leave_percent=0.1,
             mask_idx=9, coding_ids=(5, 6, 7, 8)):
    """Device-side DataCollator + DataPreprocessor (transcodon/pipeline.py:143-195, 343-361) for a
    right-padded int batch: returns (corrupted tokens int32 [B,T], labels int32 [B,T] with -100 outside the
    drawn positions, struct labels or None).  ``lens`` counts <cls> ... <eos> of each row; defaults are
    pretraining.py:178-180 and the A/U/C/G replacement set of alphabet.py:55-57."""
    if not tokens.is_cuda:
        raise _lib.HipLibraryError("mlm_mask runs on the GPU only (no CPU fallback)")
    B, T = tokens.shape
    tok = tokens.to(torch.int32).contiguous()
    ln = lens.to(device=tok.device, dtype=torch.int32).contiguous()
    out, labels = torch.empty_like(tok), torch.empty_like(tok)
    st = None if struct_labels is None else struct_labels.to(device=tok.device, dtype=torch.int32).contiguous().clone()
    ckey = (tuple(coding_ids), str(tok.device))
    coding = _CODING_CACHE.get(ckey)
    if coding is None:          # a host-to-device copy synchronises: do it once per alphabet and device
        coding = _CODING_CACHE[ckey] = torch.tensor(list(coding_ids), dtype=torch.int32, device=tok.device)
    check(_lib.load().tc_mlm_mask(tok.data_ptr(), ln.data_ptr(), out.data_ptr(), labels.data_ptr(),
                                  None if st is None else st.data_ptr(), B, T, int(seed) & (2 ** 64 - 1),
                                  mask_proportion, mask_percent, leave_percent, mask_idx, coding.data_ptr(),
                                  coding.numel(), _stream()), "tc_mlm_mask")
    return out, labels, st


def warmup_cosine(step, warmup_steps, num_steps):
    """LambdaLR factor of pretraining.py:76-82."""
    if step < warmup_steps:
        return (step + 1) / warmup_steps
    progress = (step - warmup_steps) / num_steps
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


class FlatParameters:
    """Re-homes parameters and their gradients into two flat f32 buffers, so that the optimiser
    is one kernel launch and the data-parallel gradient mean one all-reduce.

    For a ProteinBertModel the buffers use the kernels' packed layout (PackedLayout) and hold
    only the parameters its forward uses (the others never receive a gradient, so
    torch.optim.AdamW in the reference never touches them either): the backward then
    accumulates straight into ``grad`` and the GEMM weight pointers come straight out of
    ``data`` (or of its one-kernel bf16 shadow) -- no per-tensor copies, adds or concatenations."""

    def __init__(self, module_or_params):
        from .model import ProteinBertModel
        self.layout = None
        self.grads_struct = None
        if isinstance(module_or_params, ProteinBertModel):
            model = module_or_params
            params = train_parameters(model)
            if any(p.dtype != torch.float32 or not p.requires_grad for p in params):
                raise ValueError("FlatParameters needs trainable float32 parameters")
            dev = params[0].device
            self.layout = PackedLayout(model)
            self.data = torch.zeros(self.layout.total, dtype=torch.float32, device=dev)
            self.grad = torch.zeros(self.layout.total, dtype=torch.float32, device=dev)
            dviews = self.layout.views(self.data, own_second_bias=True)
            gviews = self.layout.views(self.grad, own_second_bias=True)
            self.params = params
            self.spans = {}
            with torch.no_grad():
                for p, dv, gv in zip(params, dviews, gviews):
                    dv.copy_(p.detach())
                    p.data = dv
                    p.grad = gv
                    o = (dv.data_ptr() - self.data.data_ptr()) // 4
                    self.spans[id(p)] = (o, o + p.numel())
            self._gviews = gviews
            self.grads_struct, self._grad_layers = self.layout.grads_struct(self.grad)
            self._shadow = {}
            self.grad_sync = None         # parallel.GradientSync, when gradients are exchanged
            model._flat_store = self
            _lib.weights_epoch[0] += 1
            return
        cand = list(module_or_params.parameters()) if isinstance(module_or_params, torch.nn.Module) \
            else list(module_or_params)
        seen, params = set(), []
        for p in cand:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params or any(p.dtype != torch.float32 for p in params):
            raise ValueError("FlatParameters needs float32 parameters")
        dev = params[0].device
        self.params = params
        self.spans, off = {}, 0
        for p in params:
            self.spans[id(p)] = (off, off + p.numel())
            off += _align(p.numel(), 4)
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self._gviews = []
        with torch.no_grad():
            for p in params:
                o, e = self.spans[id(p)]
                self.data[o:e].copy_(p.detach().reshape(-1))
                p.data = self.data[o:e].view(p.shape)
                self._gviews.append(self.grad[o:e].view(p.shape))
                p.grad = self._gviews[-1]

    def zero_grad(self):
        self.grad.zero_()
        for p, gv in zip(self.params, self._gviews):      # re-attach in case something replaced .grad
            if p.grad is not gv:
                p.grad = gv

    def sync_shared_bias_grad(self):
        """structure_linear.out_proj.bias and structure_linear.bias enter the forward as a sum:
        the kernels accumulate one gradient, which the second parameter's slot mirrors."""
        if self.layout is not None:
            (o, n), (o2, _) = self.layout.hoff["st_bias"], self.layout.hoff["st_bias2"]
            self.grad[o2:o2 + n].copy_(self.grad[o:o + n])

    def version_key(self):
        """Changes whenever the stored weights may have changed: the optimiser's epoch, writes
        through the flat buffer, and in-place writes through any re-homed parameter (``p.copy_``,
        ``load_state_dict``, a torch optimiser): those bump the parameter's own version counter,
        not the base buffer's."""
        return (_lib.weights_epoch[0], self.data._version, tuple(p._version for p in self.params))

    def mark_dirty(self):
        """Call after writing weights by any route the version counters cannot see."""
        _lib.weights_epoch[0] += 1

    def shadow(self, dtype):
        """``data`` in the compute dtype (itself for float32), refreshed when the weights changed."""
        if dtype == torch.float32:
            return self.data
        key = self.version_key()
        ent = self._shadow.get(dtype)
        if ent is None or ent[0] != key:
            buf = ent[1] if ent is not None else torch.empty_like(self.data, dtype=dtype)
            buf.copy_(self.data)
            self._shadow[dtype] = ent = (key, buf)
        return ent[1]

    def ranges(self, exclude=()):
        """Maximal [begin, end) element ranges of the buffers that avoid the excluded parameters."""
        cut = sorted(self.spans[id(p)] for p in exclude if id(p) in self.spans)
        out, pos = [], 0
        for b, e in cut:
            b4, e4 = b // 4 * 4, _align(e, 4)          # keep every launch 16-byte aligned
            if b4 > pos:
                out.append((pos, b4))
            pos = max(pos, e4)
        if pos < self.data.numel():
            out.append((pos, self.data.numel()))
        return out


class AdamW:
    """torch.optim.AdamW semantics on a FlatParameters store with tc_adamw_step
    (pretraining.py:64-70: lr 2e-4, weight decay 0.1 on every parameter)."""

    def __init__(self, flat, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0

    def step(self, lr_factor=1.0, grad_scale=1.0, exclude=()):
        """``exclude``: parameters without a gradient this step (torch skips those entirely,
        weight decay included), e.g. the structure head while no structure label is set."""
        self.step_count += 1
        f = self.flat
        for b, e in f.ranges(exclude):
            check(_lib.load().tc_adamw_step(f.data.data_ptr() + 4 * b, f.grad.data_ptr() + 4 * b,
                                            self.exp_avg.data_ptr() + 4 * b, self.exp_avg_sq.data_ptr() + 4 * b,
                                            e - b, self.lr * lr_factor, self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, self.step_count, grad_scale, _stream()),
                  "tc_adamw_step")
        _lib.weights_epoch[0] += 1   # parameters changed behind torch's version counters: repack


def structure_head_parameters(model):
    return list(model.structure_linear.parameters())
