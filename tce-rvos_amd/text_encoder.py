"""RoBERTa forward on the HIP kernels.

The nn.Module that OWNS the weights stays HuggingFace's RobertaModel (so `text_encoder.*` state-dict keys are the
reference's, tce_rvos.py:137); this file only reads its tensors and replays its arithmetic
(embeddings -> 12 x [self-attention, output dense + residual + LayerNorm, GELU MLP + residual + LayerNorm] -> pooler)
with tce_gemm_f32 / tce_layernorm_f32 / tce_embed_ln_f32 / tce_mha_small64_f32 / tce_tanh_f32.
HF's own forward is used only by the tests, as the checker of this restatement.
"""
import torch

from . import ops
from ._lib import check, lib
from .ops import ACT_GELU, RES_ADD, gemm_ex


class TextPlan:
    def __init__(self, hf_model):
        cfg = hf_model.config
        if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads != 64:
            raise NotImplementedError("text attention kernel is built for head_dim 64 (RoBERTa-base/large)")
        if getattr(cfg, "hidden_act", "gelu") != "gelu":
            raise NotImplementedError("only the erf GELU of RoBERTa is implemented")
        self.C, self.heads, self.ff = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        self.eps = float(cfg.layer_norm_eps)
        self.pad = int(cfg.pad_token_id)
        sd = {k: v.detach() for k, v in hf_model.state_dict().items() if v.is_floating_point()}
        for k, v in sd.items():
            if not v.is_cuda or v.dtype != torch.float32:
                raise RuntimeError(f"text_encoder.{k} must be a CUDA float32 tensor")
        self.sd = sd
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layer.{i}."
            wqkv = torch.cat([sd[p + "attention.self.query.weight"], sd[p + "attention.self.key.weight"],
                              sd[p + "attention.self.value.weight"]], 0).contiguous()
            bqkv = torch.cat([sd[p + "attention.self.query.bias"], sd[p + "attention.self.key.bias"],
                              sd[p + "attention.self.value.bias"]], 0).contiguous()
            self.layers.append((wqkv, bqkv, p))

    def forward(self, ids, A):
        """ids int64 [G, L] on the GPU (G captions of equal length: one per clip of a clip group; G = 1 for a single clip);
        A = arena allocator.  Returns (last_hidden_state [G*L, C] caption-major, pooler_output [C] for G = 1, [G, C] otherwise);
        the two live in ONE [G*L + G, C] buffer, so the resizer takes them as one tensor (pipeline.text_stage).
        (The few-row kernel of csrc/fewrow.hip was tried for the dense layers and is slower at K = 768 / 3072 than the
        split-K GEMM: 12.7 vs 9.8 us, profiles/r03_fewrow.txt.)"""
        sd, C = self.sd, self.C
        G, Ls = int(ids.shape[0]), int(ids.shape[1])
        L = G * Ls  # rows of every activation below
        s = ops._stream()
        xbuf = A(L + G, C)  # rows L.. receive the pooler outputs
        x = xbuf[:L]
        # position ids (HF create_position_ids_from_input_ids) are derived inside the kernel: no ATen arithmetic here
        emb = (sd["embeddings.word_embeddings.weight"].data_ptr(), sd["embeddings.position_embeddings.weight"].data_ptr(),
               sd["embeddings.token_type_embeddings.weight"].data_ptr(), sd["embeddings.LayerNorm.weight"].data_ptr(),
               sd["embeddings.LayerNorm.bias"].data_ptr())
        if G == 1:
            check(lib().tce_embed_ln_f32(ids.data_ptr(), None, *emb, x.data_ptr(), L, C, self.eps, self.pad, s), "tce_embed_ln_f32")
        else:
            check(lib().tce_embed_ln_seqs_f32(ids.data_ptr(), *emb, x.data_ptr(), G, Ls, C, self.eps, self.pad, s), "tce_embed_ln_seqs_f32")
        qkv, att, hdn = A(L, 3 * C), A(L, C), A(L, self.ff)
        # M = L (32 tokens) against 768..3072-deep weights: split K so that ~200 workgroups stream each weight matrix
        # instead of N/64 (ops.splitk_for); the partial sums meet in `ws`
        sk_qkv, sk_out, sk_f1, sk_f2 = (ops.splitk_for(L, n, k) for n, k in ((3 * C, C), (C, C), (self.ff, C), (C, self.ff)))
        # Weight-stream form (csrc/thin.hip): every dense layer is ONE launch that requests its whole weight matrix at once and
        # leaves K/256 partial planes; the planes meet in the next consumer (attention reads the qkv planes, fc2 reads fc1's
        # planes + bias + GELU) or in one reduce(+LayerNorm) launch: 7 launches per layer instead of 9, each a single round
        # trip (13 us -> a few us per projection, profiles/r04_thin_linear.txt)
        t_qkv, t_out, t_f1, t_f2 = (ops.thin_splits(L, n, k) for n, k in ((3 * C, C), (C, C), (self.ff, C), (C, self.ff)))
        if min(t_qkv, t_out, t_f1, t_f2) > 0:
            ws_a = A(max(t_qkv * L * 3 * C, t_f1 * L * self.ff))   # qkv planes / fc1 planes
            ws_b = A(max(t_out * L * C, t_f2 * L * C))             # out-proj planes / fc2 planes
            for wqkv, bqkv, p in self.layers:
                ops.thin_partials(x, wqkv, ws_a, L, 3 * C, C)
                check(lib().tce_mha_small64_seqs_f32(ws_a.data_ptr(), t_qkv, bqkv.data_ptr(), att.data_ptr(), G, Ls, self.heads, 0.125, s),
                      "tce_mha_small64_seqs_f32")
                ops.thin_partials(att, sd[p + "attention.output.dense.weight"], ws_b, L, C, C)
                ops.splitk_reduce(ws_b, t_out, L, C, x, bias=sd[p + "attention.output.dense.bias"], res=x, ldres=C, res_mode=RES_ADD,
                                  ln=(sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"]), eps=self.eps)
                ops.thin_partials(x, sd[p + "intermediate.dense.weight"], ws_a, L, self.ff, C)
                ops.thin_partials(ws_a, sd[p + "output.dense.weight"], ws_b, L, C, self.ff, xsplits=t_f1,
                                  bias_x=sd[p + "intermediate.dense.bias"], act_x=ACT_GELU)
                ops.splitk_reduce(ws_b, t_f2, L, C, x, bias=sd[p + "output.dense.bias"], res=x, ldres=C, res_mode=RES_ADD,
                                  ln=(sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"]), eps=self.eps)
            return self._pooler(x, xbuf, G, Ls, s)
        # (more than 128 rows -- a large clip group -- or the exact-fp32 arithmetic: tiled GEMMs; attention per caption as above)
        ws = A(max(sk_qkv * L * 3 * C, sk_out * L * C, sk_f1 * L * self.ff, sk_f2 * L * C))
        for wqkv, bqkv, p in self.layers:
            gemm_ex(x, wqkv, qkv, L, 3 * C, C, C, C, 3 * C, bias=bqkv, splitk=sk_qkv, ws=ws)
            if G == 1:
                check(lib().tce_mha_small64_f32(qkv.data_ptr(), att.data_ptr(), L, self.heads, 0.125, s), "tce_mha_small64_f32")
            else:
                check(lib().tce_mha_small64_seqs_f32(qkv.data_ptr(), 1, None, att.data_ptr(), G, Ls, self.heads, 0.125, s),
                      "tce_mha_small64_seqs_f32")
            # dense + residual + LayerNorm: the norm rides in the split-K reduction pass
            gemm_ex(att, sd[p + "attention.output.dense.weight"], x, L, C, C, C, C, C, bias=sd[p + "attention.output.dense.bias"],
                    res=x, ldres=C, res_mode=RES_ADD, splitk=sk_out, ws=ws, ln_eps=self.eps,
                    ln=(sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"]))
            gemm_ex(x, sd[p + "intermediate.dense.weight"], hdn, L, self.ff, C, C, C, self.ff,
                    bias=sd[p + "intermediate.dense.bias"], act=ACT_GELU, splitk=sk_f1, ws=ws)
            gemm_ex(hdn, sd[p + "output.dense.weight"], x, L, C, self.ff, self.ff, self.ff, C,
                    bias=sd[p + "output.dense.bias"], res=x, ldres=C, res_mode=RES_ADD, splitk=sk_f2, ws=ws, ln_eps=self.eps,
                    ln=(sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"]))
        return self._pooler(x, xbuf, G, Ls, s)

    def _pooler(self, x, xbuf, G, Ls, s):
        """pooler_output = tanh(dense(first token of each caption)) -> rows G*Ls .. of xbuf"""
        sd, C = self.sd, self.C
        pooled = xbuf[G * Ls:]
        gemm_ex(x, sd["pooler.dense.weight"], pooled, 1, C, C, C, C, C, bias=sd["pooler.dense.bias"], batch=G, sA=Ls * C, sC=C)
        check(lib().tce_tanh_f32(pooled.data_ptr(), pooled.data_ptr(), G * C, s), "tce_tanh_f32")
        return x, (pooled[0] if G == 1 else pooled)
