"""ProteinBertModel on the MI355X kernels.

Same constructor (args namespace + alphabet), ``add_args``, parameter names / state_dict
keys and forward contract as the reference's transcodon/model.py:30-263.  The forward of a
padded batch is ONE call into libtranscodon_hip.so (tc_encoder_forward); PyTorch only owns
the parameters, the output buffers and the stream.
"""

import contextlib
import ctypes as C

import torch
from torch import nn

from . import _lib, ops
from ._lib import EncoderDesc, LayerWeights, check
from .modules import (ESM1bLayerNorm, RNA_2D_RobertaLMHead, RNA_2D_Structure_Connector,
                      RobertaLMHead, TransformerLayer)
from .multihead_attention import _param_key


class ProteinBertModel(nn.Module):
    @classmethod
    def add_args(cls, parser):
        parser.add_argument("--num_layers", default=12, type=int, metavar="N", help="number of layers")
        parser.add_argument("--embed_dim", default=768, type=int, metavar="N", help="embedding dimension")
        parser.add_argument("--attention_dropout", default=0.0, type=float, help="dropout on attention")
        parser.add_argument("--logit_bias", action="store_true", help="whether to apply bias to logits")
        parser.add_argument("--rope_embedding", default=True, type=_flag, help="use rotary embeddings")
        parser.add_argument("--ffn_embed_dim", default=768 * 4, type=int, metavar="N", help="FFN width")
        parser.add_argument("--attention_heads", default=12, type=int, metavar="N", help="attention heads")
        parser.add_argument("--species_embedding", default=True, type=_flag, help="add species embedding")
        parser.add_argument("--struct_label", default=True, type=_flag, help="structure head")
        parser.add_argument("--species_vocab_size", default=1438, type=int, metavar="N",
                            help="species_vocab_size")

    def __init__(self, args, alphabet):
        super().__init__()
        self.args = args
        self.alphabet_size = len(alphabet)
        self.padding_idx = alphabet.padding_idx
        self.mask_idx = alphabet.mask_idx
        self.cls_idx = alphabet.cls_idx
        self.eos_idx = alphabet.eos_idx
        self.prepend_bos = alphabet.prepend_bos
        self.append_eos = alphabet.append_eos
        self.model_version = "ESM-1b"
        if not getattr(args, "rope_embedding", True):
            raise NotImplementedError("learned positional embeddings are not built (rope only)")
        if not getattr(args, "species_embedding", True):
            raise NotImplementedError("the species embedding is always on in this build")
        if getattr(args, "token_dropout", False) or getattr(args, "emb_layer_norm_before", False):
            raise NotImplementedError("token_dropout / emb_layer_norm_before are not built")
        if getattr(args, "attention_dropout", 0.0):
            raise NotImplementedError("attention dropout is not built (the reference trains with 0)")
        E = args.embed_dim
        self.embed_scale = 1
        self.embed_tokens = nn.Embedding(self.alphabet_size, E, padding_idx=self.padding_idx)
        self.species_embedding = nn.Embedding(args.species_vocab_size, E)
        # parameters that live in checkpoints but are never used by forward (model.py:114-134)
        self.struct_linear1 = nn.Linear(E, E)
        self.struct_linear2 = nn.Linear(E, E)
        self.struct_linear3 = nn.Linear(E, E)
        self.connector = RNA_2D_Structure_Connector(E)
        self.norm = ESM1bLayerNorm(E)
        self.existence_head = nn.Sequential(nn.Linear(E, 128), nn.ReLU(), nn.Linear(128, 1))
        self.layers = nn.ModuleList([
            TransformerLayer(E, args.ffn_embed_dim, args.attention_heads,
                             attention_dropout=0.0, add_bias_kv=False, use_esm1b_layer_norm=True,
                             rope_embedding=True)
            for _ in range(args.num_layers)])
        self.emb_layer_norm_after = ESM1bLayerNorm(E)
        self.lm_head = RobertaLMHead(embed_dim=E, output_dim=self.alphabet_size,
                                     weight=self.embed_tokens.weight)
        self.structure_linear = RNA_2D_RobertaLMHead(embed_dim=E, output_dim=3)
        self.compute_dtype = None       # None: follow the parameter dtype (f32 unless .bfloat16())
        self.check_indices = True       # raise IndexError on out-of-range token / species ids
        self.fold_layernorm = True      # bf16 inference: LayerNorms of the layers folded into the products
        self._pack = {}
        self._workspace = None
        self._train_workspace = None
        self._train_generation = 0
        self._flat_store = None         # train.FlatParameters, when the parameters live in one

    @property
    def num_layers(self):
        return self.args.num_layers

    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (MFMA bf16 operands, fp32 accumulate) or torch.float32 (exact f32)."""
        assert dtype in (None, torch.float32, torch.bfloat16)
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        return self

    def _dtype(self):
        return self.compute_dtype or (torch.bfloat16 if self.embed_tokens.weight.dtype == torch.bfloat16
                                      else torch.float32)

    # -- device-side weight pack and descriptor --------------------------------------------
    def _packed_from_store(self, dtype, T, device):
        """Descriptor whose weight pointers go straight into the flat parameter store (or its
        one-kernel shadow in the compute dtype): no per-tensor concatenation or cast."""
        store = self._flat_store
        rope_len = max(512, 1 << (T - 1).bit_length())
        key = ("store", dtype, str(device), rope_len, store.version_key())
        if self._pack.get("key") == key:
            return self._pack
        lay = store.layout
        with torch.no_grad():
            lowp = store.shadow(dtype)
            esz = lowp.element_size()
            f32 = lambda o_sz: store.data.data_ptr() + 4 * o_sz[0]
            cd = lambda o_sz: lowp.data_ptr() + esz * o_sz[0]
            keep = {"lowp": lowp,
                    "st_bias": (self.structure_linear.out_proj.bias.detach()
                                + self.structure_linear.bias.detach()).float().contiguous()}
            layers = (LayerWeights * len(self.layers))()
            per_layer = []
            for i, l in enumerate(self.layers):
                d = lay.loff[i]
                cos, sin = l.self_attn.rope.tables(rope_len, device)
                per_layer.append((cos, sin))
                for name in ("ln1_g", "ln1_b", "b_qkv", "b_o", "ln2_g", "ln2_b", "b_fc1", "b_fc2"):
                    setattr(layers[i], name, f32(d[name]))
                for name in ("w_qkv", "w_o", "w_fc1", "w_fc2"):
                    setattr(layers[i], name, cd(d[name]))
                layers[i].rope_cos, layers[i].rope_sin = cos.data_ptr(), sin.data_ptr()
            d = EncoderDesc()
            d.dtype = _lib.TC_BF16 if dtype == torch.bfloat16 else _lib.TC_F32
            d.num_layers, d.E, d.F = len(self.layers), self.args.embed_dim, self.args.ffn_embed_dim
            d.H, d.vocab = self.args.attention_heads, self.alphabet_size
            d.n_species, d.n_struct, d.pad_idx = self.args.species_vocab_size, 3, self.padding_idx
            d.rope_table_len, d.ln_eps = rope_len, self.emb_layer_norm_after.eps
            h = lay.hoff
            d.tok_table, d.species_table = f32(h["tok_table"]), f32(h["species_table"])
            d.layers = C.cast(layers, C.POINTER(LayerWeights))
            for name in ("lnf_g", "lnf_b", "b_heads", "lm_ln_g", "lm_ln_b", "lm_bias", "st_ln_g", "st_ln_b", "st_w"):
                setattr(d, name, f32(h[name]))
            d.w_heads = cd(h["w_heads"])
            d.st_bias = keep["st_bias"].data_ptr()
        self._pack = {"key": key, "desc": d, "layers": layers, "keep": keep, "per_layer": per_layer}
        return self._pack

    def _packed(self, dtype, T, device):
        if self._flat_store is not None and self._flat_store.data.device == device:
            return self._packed_from_store(dtype, T, device)
        own = [self.embed_tokens.weight, self.species_embedding.weight,
               self.emb_layer_norm_after.weight, self.emb_layer_norm_after.bias,
               self.lm_head.dense.weight, self.lm_head.dense.bias, self.lm_head.layer_norm.weight,
               self.lm_head.layer_norm.bias, self.lm_head.bias,
               self.structure_linear.dense.weight, self.structure_linear.dense.bias,
               self.structure_linear.layer_norm.weight, self.structure_linear.layer_norm.bias,
               self.structure_linear.out_proj.weight, self.structure_linear.out_proj.bias,
               self.structure_linear.bias]
        layer_params = [p for l in self.layers for p in l.parameters()]
        rope_len = max(512, 1 << (T - 1).bit_length())
        key = (dtype, str(device), rope_len, bool(self.fold_layernorm), _param_key(own + layer_params))
        if self._pack.get("key") == key:
            return self._pack
        with torch.no_grad():
            f = lambda p: p.detach().float().contiguous()
            keep = {
                "tok": f(own[0]), "sp": f(own[1]), "lnf_g": f(own[2]), "lnf_b": f(own[3]),
                "w_heads": torch.cat([own[4].detach(), own[9].detach()]).to(dtype).contiguous(),
                "b_heads": torch.cat([own[5].detach(), own[10].detach()]).float().contiguous(),
                "lm_ln_g": f(own[6]), "lm_ln_b": f(own[7]), "lm_bias": f(own[8]),
                "st_ln_g": f(own[11]), "st_ln_b": f(own[12]), "st_w": f(own[13]),
                "st_bias": (own[14].detach().float() + own[15].detach().float()).contiguous(),
            }
            layers = (LayerWeights * len(self.layers))()
            per_layer = []
            for i, l in enumerate(self.layers):
                pk = l.packed(dtype)
                cos, sin = l.self_attn.rope.tables(rope_len, device)
                per_layer.append((pk, cos, sin))
                for name in ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_g", "ln2_b",
                             "w_fc1", "b_fc1", "w_fc2", "b_fc2"):
                    setattr(layers[i], name, pk[name].data_ptr())
                layers[i].rope_cos, layers[i].rope_sin = cos.data_ptr(), sin.data_ptr()
                if dtype == torch.bfloat16 and self.fold_layernorm:
                    folded = self._fold_layer(l, pk, device)
                    per_layer.append(folded)
                    for name, t in folded.items():
                        setattr(layers[i], name, t.data_ptr())
            d = EncoderDesc()
            d.dtype = _lib.TC_BF16 if dtype == torch.bfloat16 else _lib.TC_F32
            d.num_layers, d.E, d.F = len(self.layers), self.args.embed_dim, self.args.ffn_embed_dim
            d.H, d.vocab = self.args.attention_heads, self.alphabet_size
            d.n_species, d.n_struct, d.pad_idx = self.args.species_vocab_size, 3, self.padding_idx
            d.rope_table_len, d.ln_eps = rope_len, self.emb_layer_norm_after.eps
            d.tok_table, d.species_table = keep["tok"].data_ptr(), keep["sp"].data_ptr()
            d.layers = C.cast(layers, C.POINTER(LayerWeights))
            for name in ("lnf_g", "lnf_b", "w_heads", "b_heads", "lm_ln_g", "lm_ln_b", "lm_bias",
                         "st_ln_g", "st_ln_b", "st_w", "st_bias"):
                setattr(d, name, keep[name].data_ptr())
            if dtype == torch.bfloat16 and self.fold_layernorm:
                # emb_layer_norm_after folded into the two head dense layers (model.py:232-240): used by a
                # folded forward that is not asked for the last representation
                w_heads32 = torch.cat([own[4].detach(), own[9].detach()]).float().contiguous()
                wf, sh, ch = self._fold_weights(w_heads32, keep["lnf_g"], keep["lnf_b"], keep["b_heads"], device)
                keep["w_heads_ln"], keep["s_heads_ln"], keep["c_heads_ln"] = wf, sh, ch
                for name in ("w_heads_ln", "s_heads_ln", "c_heads_ln"):
                    setattr(d, name, keep[name].data_ptr())
        self._pack = {"key": key, "desc": d, "layers": layers, "keep": keep, "per_layer": per_layer}
        return self._pack

    @staticmethod
    def _fold_weights(W, g, b, bias, device):
        """tc_ln_fold_weights on one weight matrix: (bf16(gamma o W), its row sums, beta W^T + bias)."""
        lib = _lib.load()
        N, K = W.shape
        wf = torch.empty(N, K, dtype=torch.bfloat16, device=device)
        s = torch.empty(N, dtype=torch.float32, device=device)
        c = torch.empty(N, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        check(lib.tc_ln_fold_weights(W.data_ptr(), g.data_ptr(), b.data_ptr(), bias.data_ptr(), wf.data_ptr(),
                                     s.data_ptr(), c.data_ptr(), N, K, stream), "tc_ln_fold_weights")
        torch.cuda.current_stream(device).synchronize()      # W may be freed after this
        return wf, s, c

    @staticmethod
    def _fold_layer(l, pk, device):
        """The layer's two LayerNorms folded into the q/k/v and fc1 weights (tc_ln_fold_weights):
        bf16(gamma o W), its row sums, and beta W^T + b.  With these in the descriptor and a batch
        large enough for the large-tile kernels, the bf16 forward runs no LayerNorm kernel inside the
        layers (modules.py:87,100 become part of the products on either side)."""
        lib = _lib.load()
        a = l.self_attn
        wq = torch.cat([a.q_proj.weight.detach(), a.k_proj.weight.detach(), a.v_proj.weight.detach()]).float().contiguous()
        w1 = l.fc1.weight.detach().float().contiguous()
        out = {}
        stream = torch.cuda.current_stream(device).cuda_stream
        for tag, W, g, b, bias in (("qkv", wq, pk["ln1_g"], pk["ln1_b"], pk["b_qkv"]),
                                   ("fc1", w1, pk["ln2_g"], pk["ln2_b"], pk["b_fc1"])):
            N, K = W.shape
            wf = torch.empty(N, K, dtype=torch.bfloat16, device=device)
            s = torch.empty(N, dtype=torch.float32, device=device)
            c = torch.empty(N, dtype=torch.float32, device=device)
            check(lib.tc_ln_fold_weights(W.data_ptr(), g.data_ptr(), b.data_ptr(), bias.data_ptr(), wf.data_ptr(),
                                         s.data_ptr(), c.data_ptr(), N, K, stream), "tc_ln_fold_weights")
            out[f"w_{tag}_ln"], out[f"s_{tag}_ln"], out[f"c_{tag}_ln"] = wf, s, c
        torch.cuda.current_stream(device).synchronize()      # W and the temporaries may be freed after this
        return out

    @contextlib.contextmanager
    def workspace_slot(self, k):
        """Forward calls inside the block work in workspace number ``k`` (kept between uses) instead of the
        model's default one.  The kernels of two forwards that run at the same time -- on two streams -- must
        not share a workspace: give each stream its own slot."""
        slots = self.__dict__.setdefault("_workspace_slots", {})
        saved = self._workspace
        self._workspace = slots.get(k)
        try:
            yield self
        finally:
            slots[k] = self._workspace
            self._workspace = saved

    def _get_workspace(self, nbytes, device, train=False):
        attr = "_train_workspace" if train else "_workspace"
        ws = getattr(self, attr)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            setattr(self, attr, None)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            setattr(self, attr, ws)
        return ws

    # -- forward ---------------------------------------------------------------------------
    def forward(self, tokens, species_type_ids, repr_layers=[12], need_head_weights=False, _defer_status=None):
        assert tokens.ndim == 2
        if not tokens.is_cuda:
            raise _lib.HipLibraryError(
                "ProteinBertModel.forward runs on the MI355X kernels only: move the model and "
                "its inputs to the GPU (there is no CPU fallback)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if need_head_weights:
                raise NotImplementedError("need_head_weights is an inference-only path: use torch.no_grad()")
            return self._forward_train(tokens, species_type_ids)
        if need_head_weights:
            return self._forward_with_attentions(tokens, species_type_ids, repr_layers)
        B, T = tokens.shape
        dev = tokens.device
        L, E = len(self.layers), self.args.embed_dim
        dt = self._dtype()
        pk = self._packed(dt, T, dev)
        lib = _lib.load()
        tok32 = tokens.to(torch.int32).contiguous()
        sp32 = species_type_ids.to(device=dev, dtype=torch.int32).contiguous()
        assert sp32.numel() == B
        logits = torch.empty(B, T, self.alphabet_size, dtype=torch.float32, device=dev)
        struct = torch.empty(B, T, 3, dtype=torch.float32, device=dev)
        repr_layers = set(repr_layers)
        reps = {i: torch.empty(B, T, E, dtype=torch.float32, device=dev)
                for i in sorted(repr_layers) if 0 <= i <= L}
        mids = [i for i in reps if i < L]
        rep_idx = (C.c_int32 * max(1, len(mids)))(*mids)
        rep_ptr = (C.c_void_p * max(1, len(mids)))(*[reps[i].data_ptr() for i in mids])
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        need = lib.tc_workspace_bytes(C.byref(pk["desc"]), B, T)
        ws = self._get_workspace(need, dev)
        check(lib.tc_encoder_forward(
            C.byref(pk["desc"]), tok32.data_ptr(), sp32.data_ptr(), B, T, logits.data_ptr(),
            struct.data_ptr(), reps[L].data_ptr() if L in reps else None, len(mids), rep_idx,
            rep_ptr, status.data_ptr(), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream().cuda_stream), "tc_encoder_forward")
        out = {"logits": logits, "struct_label_output": struct, "representations": reps}
        if _defer_status is not None:           # a graph capture in progress: no host read of the flag here
            _defer_status.append(status)
            return out
        if self.check_indices and int(status.item()) != 0:
            raise IndexError("token or species id out of range")
        return out

    def capture_forward(self, B, T, repr_layers=(12,), device=None, private_workspace=False):
        """The inference forward of a fixed batch shape as ONE HIP graph (see GraphedForward)."""
        return GraphedForward(self, B, T, repr_layers, device, private_workspace)

    def packed_layout(self, species_type_ids, cu_seqlens, device):
        """Device-side index arrays of a packed batch (see forward_packed): rotary position and species
        of every row, the attention work list, and the row panels of the q/k/v product (every sequence
        cut into pieces of at most 256 rows: first row, end row, position of the first row).  Depends
        only on the lengths and species, so a caller that runs the same batch repeatedly builds it once."""
        import numpy as np
        cu = np.asarray(cu_seqlens, dtype=np.int64)
        lens = np.diff(cu)
        n_seq = len(lens)
        if cu.ndim != 1 or n_seq < 1 or cu[0] != 0 or (lens <= 0).any():
            raise ValueError("packed batch: cu_seqlens must start at 0 and increase")
        sp = np.asarray(species_type_ids.cpu() if torch.is_tensor(species_type_ids) else species_type_ids, dtype=np.int64)
        if sp.shape != (n_seq,):
            raise ValueError("packed batch: one species id per sequence")
        qrows = _lib.load().tc_attn_packed_block_rows(_lib.TC_BF16 if self._dtype() == torch.bfloat16 else _lib.TC_F32)
        M = int(cu[-1])
        pos = np.arange(M, dtype=np.int64) - np.repeat(cu[:-1], lens)
        nb = (lens + qrows - 1) // qrows
        seq_of_block = np.repeat(np.arange(n_seq), nb)
        qb = np.arange(int(nb.sum())) - np.repeat(np.cumsum(nb) - nb, nb)
        order = np.argsort(-lens[seq_of_block], kind="stable")          # longest sequences first: shorter tail
        blocks = np.zeros((len(qb), 4), np.int32)
        blocks[:, 0], blocks[:, 1], blocks[:, 2] = cu[seq_of_block][order], lens[seq_of_block][order], qb[order]
        npan = (lens + 255) // 256
        seq_of_panel = np.repeat(np.arange(n_seq), npan)
        first = (np.arange(int(npan.sum())) - np.repeat(np.cumsum(npan) - npan, npan)) * 256   # position of the panel's first row
        panels = np.zeros((len(first), 4), np.int32)
        panels[:, 0] = cu[seq_of_panel] + first
        panels[:, 1] = np.minimum(cu[seq_of_panel] + first + 256, cu[seq_of_panel + 1])
        panels[:, 2] = first
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)
        return {"cu": np.ascontiguousarray(cu.astype(np.int32)), "n_seq": n_seq, "rows": M, "max_len": int(lens.max()),
                "pos": up(pos), "species_row": up(np.repeat(sp, lens)), "blocks": up(blocks), "n_blocks": len(qb),
                "panels": up(panels), "n_panels": len(first), "dtype": self._dtype()}

    def forward_packed(self, tokens, species_type_ids, cu_seqlens, want_logits=True, want_struct=True,
                       want_representation=True, layout=None):
        """Forward of a PACKED variable-length batch: ``tokens`` int [M] holds the sequences one after
        the other (no padding), sequence i in rows [cu_seqlens[i], cu_seqlens[i+1]); ``species_type_ids``
        int [n_seq]; ``cu_seqlens`` a host sequence of n_seq+1 ints starting at 0.  Every sequence is
        computed exactly as a batch-of-one forward would (model.py:175-259 has no cross-sequence term;
        rotary positions restart per sequence), but all row kernels and GEMMs see sum(T_i) rows.
        Returns {"logits" [M,V], "struct_label_output" [M,3], "representations" {L: [M,E]}}; parts not
        asked for are left out (want_logits=False skips both heads: embedding extraction).
        ``layout``: a packed_layout() result for the same lengths and species (else built here)."""
        if not tokens.is_cuda:
            raise _lib.HipLibraryError("forward_packed runs on the MI355X kernels only (there is no CPU fallback)")
        dev = tokens.device
        lay = layout if layout is not None else self.packed_layout(species_type_ids, cu_seqlens, dev)
        M, n_seq = lay["rows"], lay["n_seq"]
        if tokens.ndim != 1 or tokens.numel() != M:
            raise ValueError("forward_packed: tokens must be 1-D with cu_seqlens[-1] entries")
        L, E = len(self.layers), self.args.embed_dim
        dt = self._dtype()
        if lay["dtype"] != dt:
            raise ValueError("forward_packed: the layout was built for another compute dtype")
        pk = self._packed(dt, lay["max_len"], dev)
        lib = _lib.load()
        tok32 = tokens.to(torch.int32).contiguous()
        out = {}
        logits = struct = rep = None
        if want_logits:
            logits = out["logits"] = torch.empty(M, self.alphabet_size, dtype=torch.float32, device=dev)
            if want_struct:
                struct = out["struct_label_output"] = torch.empty(M, 3, dtype=torch.float32, device=dev)
        if want_representation or not want_logits:
            rep = torch.empty(M, E, dtype=torch.float32, device=dev)
            out["representations"] = {L: rep}
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        need = lib.tc_workspace_bytes(C.byref(pk["desc"]), 1, M)
        ws = self._get_workspace(need, dev)
        p = lambda t: None if t is None else t.data_ptr()
        check(lib.tc_encoder_forward_packed(
            C.byref(pk["desc"]), tok32.data_ptr(), lay["species_row"].data_ptr(), lay["pos"].data_ptr(),
            lay["blocks"].data_ptr(), lay["n_blocks"], p(lay.get("panels")), lay.get("n_panels", 0),
            lay["cu"].ctypes.data, n_seq, p(logits), p(struct), p(rep),
            status.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
            "tc_encoder_forward_packed")
        if self.check_indices and int(status.item()) != 0:
            raise IndexError("token or species id out of range")
        return out

    def _forward_train(self, tokens, species_type_ids):
        """Differentiable forward (the training_step call of pretraining.py:103): logits and
        structure logits carry the autograd graph; no representations are returned."""
        from .train import EncoderTrainFunction, train_parameters
        tok32 = tokens.to(torch.int32).contiguous()
        sp32 = species_type_ids.to(device=tokens.device, dtype=torch.int32).contiguous()
        assert sp32.numel() == tokens.shape[0]
        logits, struct = EncoderTrainFunction.apply(self, tok32, sp32, *train_parameters(self))
        return {"logits": logits, "struct_label_output": struct, "representations": {}}

    def _forward_with_attentions(self, tokens, species_type_ids, repr_layers):
        """Layer-by-layer path for need_head_weights=True (model.py:213-256): also returns
        ``attentions`` [B, L, H, T, T] = pre-softmax scores times the pad outer product."""
        B, T = tokens.shape
        L, E = len(self.layers), self.args.embed_dim
        x, pad, bad = ops.embed(tokens, species_type_ids.to(tokens.device), self.embed_tokens.weight.detach().float(),
                                self.species_embedding.weight.detach().float(), self.padding_idx)
        if self.check_indices and int(bad.item()) != 0:
            raise IndexError("token or species id out of range")
        repr_layers = set(repr_layers)
        reps = {}
        x = x.view(B, T, E)
        if 0 in repr_layers:
            reps[0] = x
        padding_mask = pad.view(B, T).bool()
        kpm = padding_mask if bool(padding_mask.any()) else None
        x = x.transpose(0, 1)
        attn = []
        for i, layer in enumerate(self.layers):
            x, a = layer(x, self_attn_padding_mask=kpm, need_head_weights=True)
            if (i + 1) in repr_layers:
                reps[i + 1] = x.transpose(0, 1)
            attn.append(a.transpose(1, 0))
        x = self.emb_layer_norm_after(x).transpose(0, 1).contiguous()
        if L in repr_layers:
            reps[L] = x
        out = {"logits": self.lm_head(x), "struct_label_output": self.structure_linear(x),
               "representations": reps}
        attentions = torch.stack(attn, 1)
        if kpm is not None:
            m = 1 - kpm.type_as(attentions)
            m = m.unsqueeze(1) * m.unsqueeze(2)
            attentions = attentions * m[:, None, None, :, :]
        out["attentions"] = attentions
        return out


def _flag(v):
    """Boolean command-line value that can actually be switched off."""
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() not in ("0", "false", "no", "off", "")


class GraphedForward:
    """``ProteinBertModel.forward`` (model.py:175-259, inference, no attention weights) for one fixed
    batch shape, captured into a HIP graph: the ~150 kernel launches of a forward become one graph
    launch, which is what a small batch's latency consists of.  Calling it copies the inputs into the
    graph's static buffers, replays the graph and returns the same dict as ``forward``; the tensors in
    it are the graph's own output buffers and are overwritten by the next call (clone what must live
    longer).

    The graph addresses the model's PACKED weights (concatenated q/k/v, compute-dtype casts, folded
    LayerNorm products) and a workspace by raw device pointers.  It therefore keeps a reference to the
    pack and to the workspace it was captured with (neither can be freed under it), remembers the
    version of every parameter, and before each replay compares that with the model's current state:
    after ``load_state_dict``, an optimiser step, ``set_compute_dtype`` or a move of the parameters the
    graph is captured again (three eager forwards), so a replay never runs on stale or recycled
    memory.  ``check_weights = False`` skips the comparison (about 0.1 ms of host time per call) for
    callers that promise not to touch the weights.

    The kernels of a graph work in the model's workspace, so two graphs of one model may not run at the same
    time -- unless they are captured with ``private_workspace=True``: such a graph owns its workspace, and two
    of them replayed on two streams overlap (the tail of one forward's kernels with the next one's)."""

    def __init__(self, model, B, T, repr_layers=(12,), device=None, private_workspace=False):
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("GraphedForward needs the model on the GPU (there is no CPU fallback)")
        self.model, self.B, self.T = model, int(B), int(T)
        self.device = dev
        self.tokens = torch.full((self.B, self.T), model.padding_idx, dtype=torch.int32, device=dev)
        self.tokens[:, 0] = model.cls_idx
        self.species = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.repr_layers = list(repr_layers)
        self.private_workspace = bool(private_workspace)
        self.check_weights = True
        self.captures = 0
        self._workspace = None
        self._capture()

    def _weights_state(self):
        """Cheap fingerprint of everything the captured pointers depend on."""
        m = self.model
        store = m._flat_store
        if store is not None and store.data.device == self.device:
            return ("store", m._dtype(), bool(m.fold_layernorm), store.data.data_ptr(), store.version_key())
        return (m._dtype(), bool(m.fold_layernorm), _lib.weights_epoch[0],
                tuple((p.data_ptr(), p._version) for p in m.parameters()))

    def _capture(self):
        model, dev = self.model, self.device
        shared_ws = model._workspace
        if self.private_workspace:
            model._workspace = self._workspace      # None the first time: the calls below allocate it
        try:
            with torch.no_grad():
                # first calls outside the capture: weight packing, workspace, one-time kernel attributes
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        model(self.tokens, self.species, repr_layers=self.repr_layers)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                deferred = []
                # thread-local capture mode: other threads of the process (a collective library's watchdog, a data
                # loader) may call into the runtime while this thread records
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = model(self.tokens, self.species, repr_layers=self.repr_layers, _defer_status=deferred)
            # what the recorded kernels address: held here so that no later repack, reload or workspace
            # regrowth of the model can hand that memory to someone else while this graph can still replay
            self.graph, self.out, self.status = graph, out, deferred[0]
            self._pack = model._pack
            self._workspace = model._workspace
            self._state = self._weights_state()
            self.captures += 1
        finally:
            if self.private_workspace:
                model._workspace = shared_ws

    def __call__(self, tokens, species_type_ids, check_indices=None):
        if tuple(tokens.shape) != (self.B, self.T):
            raise ValueError(f"GraphedForward was captured for tokens of shape {(self.B, self.T)}, got {tuple(tokens.shape)}")
        if self.check_weights and self._weights_state() != self._state:
            self._capture()                         # weights, dtype or fold changed since the capture
        self.tokens.copy_(tokens)
        self.species.copy_(species_type_ids.reshape(-1))
        self.graph.replay()
        chk = self.model.check_indices if check_indices is None else check_indices
        if chk and int(self.status.item()) != 0:
            raise IndexError("token or species id out of range")
        return self.out
