"""torch.autograd.Function wrappers over the predictor kernels (include/emloco_predictor.h).

PyTorch carries the autograd graph, device memory and the current stream; every contraction, softmax and
layer norm below runs in libemloco_hip.so.  There is no fallback: without the library / a GPU these raise.
"""
import ctypes as C
import os

import torch

from .. import _lib as L
from ..sim import current_stream_handle

GEMM_BIAS, GEMM_RELU, GEMM_ACC, GEMM_DROPOUT = 1, 2, 4, 8
_bound = False


class LocoValStep(C.Structure):
    """EmlocoLocoValStep (include/emloco_predictor.h)."""
    _fields_ = [("n_env", C.c_int32), ("step_to_pred", C.c_int32), ("gamma", C.c_float), ("inversion_penalty", C.c_float),
                ("min_cum_rewards", C.c_float), ("max_cum_rewards", C.c_float),
                ("current_rewards", C.c_void_p), ("current_lengths", C.c_void_p), ("current_combined_rewards", C.c_void_p),
                ("discount_coefs", C.c_void_p), ("waypoint_traj", C.c_void_p), ("init_pose", C.c_void_p), ("init_vel", C.c_void_p),
                ("traj13", C.c_void_p), ("pose", C.c_void_p), ("vel", C.c_void_p), ("target", C.c_void_p), ("weight", C.c_void_p),
                ("staged_reward", C.c_void_p), ("staged_done", C.c_void_p)]          # staged mode (both NULL: off)


def _lib():
    global _bound
    lib = L.require_device()
    if not _bound:
        vp, ci, cf, cl = C.c_void_p, C.c_int, C.c_float, C.c_int64
        lib.emloco_gemm_f32.argtypes = [ci, ci, ci, ci, cf, vp, ci, cl, ci, vp, ci, cl, ci, vp, ci, cl, vp, ci, ci, vp, vp]
        lib.emloco_gemm_f32_ex.argtypes = [ci, ci, ci, ci, cf, vp, ci, cl, ci, vp, ci, cl, ci, vp, ci, cl, vp, ci, ci, vp, cf, C.c_uint32, vp]
        lib.emloco_act_bwd.argtypes = [cl, vp, vp, ci, cf, C.c_uint32, vp, vp]
        lib.emloco_act_bwd_colsum.argtypes = [ci, ci, vp, vp, ci, cf, C.c_uint32, vp, vp, vp, vp]
        lib.emloco_softmax_fwd.argtypes = [ci, ci, ci, cf, vp, vp, vp, vp]
        lib.emloco_softmax_bwd.argtypes = [ci, ci, cf, vp, vp, vp, vp]
        lib.emloco_layernorm_fwd.argtypes = [ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_layernorm_fwd_save.argtypes = [ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_layernorm_bwd.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_layernorm_bwd2.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_colsum.argtypes = [ci, ci, vp, vp, vp, vp]
        lib.emloco_colsum_ex.argtypes = [ci, ci, vp, vp, vp, ci, vp]
        lib.emloco_attention_fwd.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, vp]
        lib.emloco_attention_bwd.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_attention_fwd_ex.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, ci, vp]
        lib.emloco_attention_bwd_ex.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, ci, vp]
        lib.emloco_gemm_relu_bwd_workspace.argtypes = [ci, ci]
        lib.emloco_gemm_relu_bwd_workspace.restype = C.c_int64
        lib.emloco_gemm_relu_bwd.argtypes = [ci, ci, ci, vp, ci, vp, ci, ci, vp, vp, cf, vp, vp, ci, vp]
        lib.emloco_attention_fwd_queries.argtypes = [ci, ci, ci, ci, ci, cf, vp, vp, vp, vp, ci, cf, C.c_uint32, vp]
        lib.emloco_attention_bwd_queries.argtypes = [ci, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, ci, cf, C.c_uint32, vp]
        lib.emloco_attention_keep_mask.argtypes = [C.c_uint32, ci, ci, cf, vp]
        lib.emloco_dropout_keep_mask.argtypes = [C.c_uint32, C.c_uint64, C.c_int64, cf, vp]
        lib.emloco_attention_fwd_dropout.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, ci, cf, C.c_uint32, vp]
        lib.emloco_attention_bwd_dropout.argtypes = [ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp, vp, ci, cf, C.c_uint32, vp]
        lib.emloco_colsum_workspace.argtypes = [ci, ci]
        lib.emloco_colsum_workspace.restype = C.c_int64
        lib.emloco_layernorm_bwd_workspace.argtypes = [ci, ci]
        lib.emloco_layernorm_bwd_workspace.restype = C.c_int64
        lib.emloco_locoval_fwd.argtypes = [ci, vp, ci] + [vp] * 14
        lib.emloco_locoval_fwd_rows.argtypes = [ci, vp, ci] + [vp] * 15
        lib.emloco_locoval_bwd.argtypes = [ci, vp, ci] + [vp] * 15
        lib.emloco_locoval_bwd_workspace.argtypes = [ci]
        lib.emloco_locoval_bwd_workspace.restype = C.c_int64
        lib.emloco_locoval_returns.argtypes = [C.POINTER(LocoValStep), vp, vp, vp, vp, vp]
        lib.emloco_locoval_returns_finish.argtypes = [C.POINTER(LocoValStep), vp, vp]
        lib.emloco_locoval_fit_grad.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp]
        lib.emloco_locoval_bwd_rows.argtypes = [ci, vp, ci] + [vp] * 17
        lib.emloco_adamw_gated.argtypes = [ci] + [vp] * 7 + [cf] * 5 + [vp, vp]
        lib.emloco_adam_clip_flat.argtypes = [C.c_int64] + [vp] * 4 + [cf, C.c_double, C.c_double] + [cf] * 5 + [vp, vp]
        lib.emloco_adam_clip_flat_counted.argtypes = [C.c_int64] + [vp] * 4 + [cf, C.c_double, C.c_double] + [cf] * 3 + [vp, vp, vp]
        lib.emloco_gather_flat.argtypes = [ci, vp, vp, vp, vp, vp]
        lib.emloco_adam_clip_flat_workspace.argtypes = [C.c_int64]
        lib.emloco_gemm_split_image_words.argtypes = [C.c_int, C.c_int]
        lib.emloco_gemm_split_image_words.restype = C.c_int64
        lib.emloco_gemm_split_pack.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        lib.emloco_adam_clip_flat_workspace.restype = C.c_int64
        lib.emloco_gemm_enable_timing.argtypes = [ci]
        lib.emloco_ffn_fwd.argtypes = [ci, ci] + [vp] * 8 + [cf, C.c_uint32, C.c_uint32, vp]
        lib.emloco_ffn_bwd_input.argtypes = [ci, ci] + [vp] * 6 + [cf, vp]
        lib.emloco_ffn_bwd_input_colsum.argtypes = [ci, ci] + [vp] * 6 + [cf, vp, vp]
        lib.emloco_ffn_fwd_norm.argtypes = [ci, ci] + [vp] * 10 + [cf] + [vp] * 4 + [cf, C.c_uint32, C.c_uint32, vp]
        lib.emloco_ffn_bwd_colsum_rows.argtypes = [ci]
        lib.emloco_ffn_bwd_colsum_rows.restype = C.c_int64
        lib.emloco_ffn_keep_mask.argtypes = [C.c_uint32, cl, cl, ci, cf, vp]
        lib.emloco_disc_reward.argtypes = [ci, vp, cf, vp, vp]
        lib.emloco_gemm_timing_stats.argtypes = [C.POINTER(ci), C.POINTER(cf), C.POINTER(C.c_double)]
        lib.emloco_gemm_timing_bytes.argtypes = [C.POINTER(C.c_double)]
        _bound = True
    return lib


def _p(t, offset=0):
    """device address of element `offset` of a tensor (fp32, or bf16 in the reduced-precision mode)"""
    return None if t is None else C.c_void_p(t.data_ptr() + t.element_size() * offset)


def _st(t):
    return current_stream_handle(t.device)


def _chk(rc, what):
    if rc != 0:
        raise L.EmlocoError(f"{what} failed with code {rc}")


GEMM_BF16 = 16
GEMM_SPLIT = 1024      # EMLOCO_GEMM_SPLIT: fp32-class products from bf16 pieces (six bf16 matrix instructions per 16 k)
GEMM_A16, GEMM_B16, GEMM_C16, GEMM_MASK16 = 64, 128, 256, 512      # EMLOCO_GEMM_*_BF16MEM: that operand is bf16 in memory
GEMM_B_SPLITIMG = 2048  # EMLOCO_GEMM_B_SPLITIMG: B is the piece image of a weight (split_image)
GEMM_SPLIT2 = 4096      # EMLOCO_GEMM_SPLIT2: with GEMM_SPLIT, two pieces per operand (three piece products): the backward's gradient products
# Pieces per operand of the split mode's BACKWARD products (round 6; the forward always runs on three: logits to 5e-7 of the reference).
# Two pieces = the three piece products above 2^-16 of a product instead of the six above 2^-24: half the matrix instructions and a shorter
# cut, on a SIMD whose matrix and vector pipes take turns (profiles/r06_attn_dq_ablation.txt).  "dx": the input-gradient GEMMs (dx = dy W,
# the fused ReLU backward; compute-bound at K = 1024: 170 TFLOP/s); "dw": the weight gradients (dW = dy^T x).  The fused attention's
# backward has its own switch inside the library (EMLOCO_ATTN_BWD_PIECES).  Measured on one box (profiles/r06_ab_bwd_pieces.txt): fp32-class
# JTA train step 138 ms (all three) -> 127.5 (attention) -> 123 (+ dx) -> 119 (+ dw); gradients of the shipped-depth model against the
# reference's: 4-9e-6 of a tensor's scale next to the loss (6e-7 with three pieces; the tests' bar is 2e-4), unchanged in the early
# layers (3e-5 .. 1.3e-4: the ReLU / LayerNorm stack's own fp32 noise, bar 1e-3).  EMLOCO_BWD_PIECES=3 (and EMLOCO_ATTN_BWD_PIECES=3)
# restore round 5's backward.
_BWD_PIECES = {"dx": int(os.environ.get("EMLOCO_BWD_PIECES_DX", os.environ.get("EMLOCO_BWD_PIECES", "2"))),
               "dw": int(os.environ.get("EMLOCO_BWD_PIECES_DW", os.environ.get("EMLOCO_BWD_PIECES", "2")))}


_FFN_COLSUM = os.environ.get("EMLOCO_FFN_COLSUM", "1") != "0"      # the chained feed-forward's backward sums dz1's columns itself (round 6)


def backward_pieces():
    """{"dx": 2 | 3, "dw": 2 | 3}: pieces per operand of the split mode's gradient GEMMs (see above)."""
    return dict(_BWD_PIECES)


def _bwd_flags(kind):
    return GEMM_SPLIT2 if _BWD_PIECES[kind] == 2 else 0
# Frozen weights (the rollout's policy and discriminator) are cut into their bf16 pieces ONCE, when first used (learning/policy_runner.py:
# +1.4 % on the policy forward, +1.8 % on the configs[2] loop, profiles/r05_ab_weight_image.txt); EMLOCO_GEMM_WEIGHT_IMAGE=0: never.
# A trained weight would have to be cut once per launch (one small extra launch): measured on the predictor's tall GEMMs that buys
# nothing (40 % fewer vector instructions in the main loop, the same 132 ms step: those launches are bound by their epilogues and by
# HBM, not by the cut), so it is off unless EMLOCO_GEMM_WEIGHT_IMAGE_ROWS names a row count from which to do it (the bit-equality
# test does).
_IMAGE_FROZEN = os.environ.get("EMLOCO_GEMM_WEIGHT_IMAGE", "1") != "0"
_IMAGE_MIN_ROWS = int(os.environ.get("EMLOCO_GEMM_WEIGHT_IMAGE_ROWS", "0")) or (1 << 62)


def split_image(W, n, k, ld, trans):
    """Piece image (emloco_gemm_split_pack) of the B operand B(n, k) = W[n * ld + k] (trans = 0) / W[k * ld + n] (trans = 1)."""
    lib = _lib()
    img = torch.empty(lib.emloco_gemm_split_image_words(int(n), int(k)), dtype=torch.int32, device=W.device)
    _chk(lib.emloco_gemm_split_pack(_p(W), int(n), int(k), int(ld), int(trans), _p(img), _st(W)), "emloco_gemm_split_pack")
    return img


def _weight_image(W, rows, n, k, ld, trans):
    """The image for a launch of `rows` output rows, or None where it does not pay / does not apply."""
    if rows < _IMAGE_MIN_ROWS or _matmul_precision[0] != "fp32_split" or W.dtype != torch.float32 or n <= 32:
        return None
    return split_image(W, n, k, ld, trans)
ATTN_BF16 = 16
ATTN_SPLIT = 64        # EMLOCO_ATTN_SPLIT: the fused attention's tile products as fp32-class sums of bf16 piece products
ATTN_QKV16 = 32        # EMLOCO_ATTN_QKV_BF16MEM
# The default is the split mode: every fp32 operand is cut into three bf16 pieces (8 + 8 + 8 mantissa bits, exact) and the six
# piece products that carry more than 2^-24 of the result run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Against float64
# its error is at or below the fp32 matrix instruction's on every shape of the train step and the policy (1.7e-7 vs 2.6e-7 of
# sum |a||b| on the q|k|v projection, profiles/r03_gemm_split.txt), the 1e-4 parity tests hold in it, and it is 1.25-1.37x faster
# (the bf16 pipe is 16x the fp32 one; the split costs 6 instructions where fp32 needs 8).  EMLOCO_MATMUL_PRECISION=fp32 puts
# every GEMM back on v_mfma_f32_32x32x2_f32.
DEFAULT_PRECISION = os.environ.get("EMLOCO_MATMUL_PRECISION", "fp32_split")       # "fp32" | "fp32_split" | "bf16"
_ATTN_SPLIT = os.environ.get("EMLOCO_ATTN_SPLIT", "1") != "0"        # (A/B knob: the fused attention of the split mode on the fp32 matrix instruction)
_matmul_precision = [DEFAULT_PRECISION]


def set_matmul_precision(mode):
    """"fp32_split" (default: fp32-class products rebuilt from bf16 pieces, see DEFAULT_PRECISION), "fp32" (fp32 operands on the
    fp32 matrix instruction) -- the 1e-4 parity tests hold in both -- or "bf16"
    (operands rounded to bf16 on their way into the matrix cores, fp32 accumulation; ~1e-3 relative output error) for
    every `linear` / projection GEMM and every fused attention launched afterwards; the two large activations of an encoder
    layer -- the feed-forward hidden layer (M x 1024) and the fused q|k|v projection (M x 384) -- and their gradients are then
    kept in HBM as bf16 (softmax statistics, LayerNorm, residual stream, weight gradients, losses and the optimiser stay fp32)."""
    if mode not in ("fp32", "fp32_split", "bf16"):
        raise ValueError("matmul precision must be 'fp32', 'fp32_split' or 'bf16'")
    _matmul_precision[0] = mode


def get_matmul_precision():
    return _matmul_precision[0]


def gemm(batch, m, n, k, A, lda, sa, ta, B, ldb, sb, tb, Cm, ldc, sc, alpha=1.0, bias=None, flags=0, ksplit=1,
         a_off=0, b_off=0, c_off=0, drop_p=0.0, drop_seed=0, b_image=None):
    """Raw strided batched GEMM: C_b[m][n] (+)= alpha * sum_k A_b(m,k) B_b(n,k) (see the header for the layouts).  bf16 tensors
    (the reduced-precision mode's large activations) are passed as they are: the dtype travels in the flags.
    b_image: the piece image of B (`split_image`) -- the split mode then reads it instead of B (same bits)."""
    if _matmul_precision[0] == "bf16":
        flags |= GEMM_BF16
    elif _matmul_precision[0] == "fp32_split":
        flags |= GEMM_SPLIT
        if b_image is not None and batch == 1 and not ta and A.dtype == torch.float32 and n > 32 and lda % 4 == 0 and (A.data_ptr() + 4 * a_off) % 16 == 0:
            B, b_off, flags = b_image, 0, flags | GEMM_B_SPLITIMG
    for t, bit in ((A, GEMM_A16), (B, GEMM_B16), (Cm, GEMM_C16)):
        if t.dtype == torch.bfloat16:
            assert a_off == 0 and b_off == 0 and c_off == 0
            flags |= bit | GEMM_BF16
    ws = None
    if ksplit > 1:
        ws = torch.empty(ksplit * batch * m * n, dtype=torch.float32, device=Cm.device)
    if drop_p > 0.0:
        flags |= GEMM_DROPOUT
    rc = _lib().emloco_gemm_f32_ex(batch, m, n, k, float(alpha), _p(A, a_off), lda, sa, ta, _p(B, b_off), ldb, sb, tb,
                                   _p(Cm, c_off), ldc, sc, _p(bias), flags, ksplit, _p(ws), float(drop_p), int(drop_seed) & 0xFFFFFFFF,
                                   _st(Cm))
    _chk(rc, "emloco_gemm_f32_ex")


_drop_counter = [0]


def next_dropout_seed():
    """32-bit seed of the next fused-dropout launch: a host-side counter mixed with torch's seed (no device sync; runs
    repeat under torch.manual_seed).  The mask itself is a stateless hash of (seed, element index) on the device."""
    _drop_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _drop_counter[0] * 0x85EBCA6B) & 0xFFFFFFFF


def _ksplit_for(red, out_elems):
    """Split a long reduction so the launch has enough workgroups (>= ~512) without a huge workspace."""
    tiles = max(1, (out_elems + 128 * 128 - 1) // (128 * 128))
    want = max(1, 512 // tiles)
    return int(max(1, min(want, red // 256, 512)))


def colsum(X2d):
    m, n = X2d.shape
    out = torch.empty(n, dtype=torch.float32, device=X2d.device)
    ws = torch.empty(_lib().emloco_colsum_workspace(m, n), dtype=torch.float32, device=X2d.device)
    _chk(_lib().emloco_colsum_ex(m, n, _p(X2d), _p(out), _p(ws), GEMM_A16 if X2d.dtype == torch.bfloat16 else 0, _st(X2d)), "emloco_colsum")
    return out


class LinearFn(torch.autograd.Function):
    """y = dropout(relu?(x W^T + b)); x (..., K), W (N, K).  Bias, ReLU and inverted dropout live in the GEMM epilogue; the
    backward applies both masks in one pass (`emloco_act_bwd`) before the two gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, W, b, relu, drop_p=0.0, drop_seed=0, out_bf16=False):
        xs = x.shape
        x2 = x.contiguous().view(-1, xs[-1])
        M, K = x2.shape
        N = W.shape[0]
        Wc = W.contiguous()
        # out_bf16: the result lives in HBM as bf16 (the fused q|k|v projection in the reduced-precision mode); its gradient
        # arrives as bf16 too and is consumed as it is by the two gradient GEMMs and the bias-gradient column sum
        y = torch.empty((M, N), dtype=torch.bfloat16 if (out_bf16 and _matmul_precision[0] == "bf16") else torch.float32, device=x.device)
        flags = (GEMM_BIAS if b is not None else 0) | (GEMM_RELU if relu else 0)
        # (split-K for the learners' small batches: a 2 048 x 1 024 x 2 048 layer is 128 tiles on 256 CUs -- 124 us whole, 62 us in
        # four k-slices, tools/exp/probe_small_gemm.py; the predictor's tall activations have thousands of tiles and stay whole)
        gemm(1, M, N, K, x2, K, 0, 0, Wc, K, 0, 0, y, N, 0, bias=b.contiguous() if b is not None else None, flags=flags,
             drop_p=drop_p, drop_seed=drop_seed, ksplit=1 if y.dtype == torch.bfloat16 else _ksplit_for(K, M * N),
             b_image=_weight_image(Wc, M, N, K, K, 0) if x2.dtype == torch.float32 and y.dtype == torch.float32 else None)
        ctx.save_for_backward(x2, Wc, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.xs, ctx.drop = relu, b is not None, xs, (float(drop_p), int(drop_seed))
        ctx.w_param = W if getattr(W, "_emloco_direct_grad", False) else None     # (see `mark_direct_grad`)
        return y.view(*xs[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        dy2 = dy.contiguous().view(M, N)
        p, seed = ctx.drop
        dx = dW = db = None
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.relu or p > 0.0:
            dz = torch.empty_like(dy2)
            if need_db:                                  # masks and bias gradient in one pass over dy
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
                ws = torch.empty(_lib().emloco_colsum_workspace(M, N), dtype=torch.float32, device=dy.device)
                _chk(_lib().emloco_act_bwd_colsum(M, N, _p(dy2), _p(y), 1 if ctx.relu else 0, p, seed & 0xFFFFFFFF, _p(dz), _p(db),
                                                  _p(ws), _st(dy2)), "emloco_act_bwd_colsum")
            else:
                _chk(_lib().emloco_act_bwd(M * N, _p(dy2), _p(y), 1 if ctx.relu else 0, p, seed & 0xFFFFFFFF, _p(dz), _st(dy2)), "emloco_act_bwd")
            dy2 = dz
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            gemm(1, M, K, N, dy2, N, 0, 0, W, K, 0, 1, dx, K, 0, ksplit=1 if dy2.dtype == torch.bfloat16 else _ksplit_for(N, M * K),
                 b_image=_weight_image(W, M, K, N, K, 1) if (dy2.dtype == torch.float32 and _BWD_PIECES["dx"] == 3) else None,
                 flags=_bwd_flags("dx"))                                                                                                # dx = dy W
            dx = dx.view(ctx.xs)
        if ctx.needs_input_grad[1]:
            g = ctx.w_param.grad if ctx.w_param is not None else None
            if g is not None and g.dtype == torch.float32 and g.shape == (N, K) and g.is_contiguous() and g.data_ptr() % 16 == 0 and dy2.dtype == torch.float32:
                # the weight gradient goes straight INTO the parameter's .grad (a view of the learner's flat bucket): C += dy^T x in the
                # GEMM's epilogue instead of a fresh N x K tensor and autograd's accumulation launch behind it
                gemm(1, N, K, M, dy2, N, 0, 1, x2, K, 0, 1, g, K, 0, ksplit=_ksplit_for(M, N * K), flags=GEMM_ACC | _bwd_flags("dw"))
            else:
                dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
                gemm(1, N, K, M, dy2, N, 0, 1, x2, K, 0, 1, dW, K, 0, ksplit=_ksplit_for(M, N * K), flags=_bwd_flags("dw"))   # dW = dy^T x
        if need_db and db is None:
            db = colsum(dy2)
        return dx, dW, db, None, None, None, None


class ReluMaskFn(torch.autograd.Function):
    """t * [h > 0] in ONE launch each way (`emloco_act_bwd` used as a forward op): the ReLU derivative of the explicit gradient network of
    the discriminator's gradient penalty (learning/amp_agent.py), which torch writes as compare + cast + multiply.  No gradient reaches h
    (the mask is piecewise constant), as with the torch expression."""

    @staticmethod
    def forward(ctx, t, h):
        t2, h2 = t.contiguous(), h.contiguous()
        out = torch.empty_like(t2)
        _chk(_lib().emloco_act_bwd(t2.numel(), _p(t2), _p(h2), 1, 0.0, 0, _p(out), _st(t2)), "emloco_act_bwd")
        ctx.save_for_backward(h2)
        return out

    @staticmethod
    def backward(ctx, dy):
        (h2,) = ctx.saved_tensors
        d2 = dy.contiguous()
        out = torch.empty_like(d2)
        _chk(_lib().emloco_act_bwd(d2.numel(), _p(d2), _p(h2), 1, 0.0, 0, _p(out), _st(d2)), "emloco_act_bwd")
        return out, None


def relu_mask(t, h):
    """t where h > 0, else 0 (same shapes, fp32, CUDA: one launch; otherwise the torch expression)."""
    if t.is_cuda and t.dtype == torch.float32 and h.dtype == torch.float32 and t.shape == h.shape:
        return ReluMaskFn.apply(t, h)
    return (h > 0).to(t.dtype) * t


def mark_direct_grad(params, on=True):
    """Weights whose gradient `LinearFn.backward` may accumulate directly into `.grad` (GEMM epilogue, C += ...) instead of returning it
    to autograd.  The caller vouches that (1) `.grad` exists and is zeroed before every backward (FlatGradBucket) and (2) every use of
    the weight in one backward pass runs on ONE stream (the accumulations are then stream-ordered; autograd's own accumulation node
    is what orders gradients that arrive from several streams)."""
    for p_ in params:
        p_._emloco_direct_grad = bool(on)


def linear(x, W, b=None, relu=False, drop_p=0.0, out_bf16=False):
    """nn.Linear (+ReLU) (+nn.Dropout(p) in training: pass drop_p > 0) as one GEMM launch."""
    K = x.shape[-1]
    if K % 4 and K >= 256:
        # a long reduction whose rows are not 16-byte aligned (the AMP discriminator's 3 090 inputs) falls off the 16-byte-load /
        # split-mode kernels onto the scalar-load fp32 ones (measured: 59 TFLOP/s).  Zero-padding both operands to a multiple of 4
        # costs two small copies and puts the three GEMMs of the layer (forward, input gradient, weight gradient) on the fast path;
        # autograd slices the gradients back.
        pad = 4 - K % 4
        x = torch.nn.functional.pad(x, (0, pad))
        W = torch.nn.functional.pad(W, (0, pad))
    N = W.shape[0]
    if N % 4 and N >= 32 and x.numel() // x.shape[-1] >= 1024:
        # the same for an output width that is not a multiple of 4 (the policy's 69 actions, the 3 090-wide input gradient of the
        # discriminator's gradient penalty): the incoming gradient is then a 16-byte-aligned operand of the two backward GEMMs
        padn = 4 - N % 4
        W = torch.nn.functional.pad(W, (0, 0, 0, padn))
        b = torch.nn.functional.pad(b, (0, padn)) if b is not None else None
        return linear(x, W, b, relu, drop_p, out_bf16)[..., :N]
    if drop_p > 0.0:
        return LinearFn.apply(x, W, b, relu, float(drop_p), next_dropout_seed(), out_bf16)
    return LinearFn.apply(x, W, b, relu, 0.0, 0, out_bf16)




# The chained feed-forward kernels (csrc/ffn_kernels.hip) serve the reduced-precision mode at the model's width; EMLOCO_FFN_CHAIN=0
# puts the block back on the separate GEMMs (A/B knob)
_FFN_CHAIN = os.environ.get("EMLOCO_FFN_CHAIN", "1") != "0"


def _ffn_chain_ok(M, K, F, N):
    return _FFN_CHAIN and _matmul_precision[0] == "bf16" and K == 128 and N == 128 and 64 <= F <= 2048 and F % 64 == 0


class FeedForwardFn(torch.autograd.Function):
    """f = dropout(linear2(dropout(relu(linear1(x))))) -- the feed-forward block of nn.TransformerEncoderLayer
    (model_jta.py:177) as one autograd node, so that the backward can fuse across the two layers: the gradient w.r.t. the
    hidden activations is masked (ReLU and dropout: hidden > 0) and column-summed (bias gradient) in the epilogue of the GEMM
    that produces it (`emloco_gemm_relu_bwd`); the unmasked M x ff gradient never exists in memory.

    Reduced-precision mode at the model's width (d = 128, ff a multiple of 64): the two products of the forward -- and the two of the
    input-gradient pass -- are CHAINED in one launch each (`emloco_ffn_fwd`, `emloco_ffn_bwd_input`, csrc/ffn_kernels.hip): the hidden
    tile goes from one product to the next in registers, the hidden layer and its gradient cross HBM four times per layer (as bf16)
    instead of seven."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, drop_p, seed1, seed2):
        xs = x.shape
        x2 = x.contiguous().view(-1, xs[-1])
        M, K = x2.shape
        F, N = W1.shape[0], W2.shape[0]
        W1c, W2c = W1.contiguous(), W2.contiguous()
        ctx.xs, ctx.drop = xs, (float(drop_p), int(seed1), int(seed2))
        # (the chained kernels read whole 16-byte groups of x: a view whose storage offset is not a multiple of four floats goes to the
        # separate GEMMs instead of failing in emloco_ffn_fwd)
        ctx.chain = _ffn_chain_ok(M, K, F, N) and x2.dtype == torch.float32 and x2.data_ptr() % 16 == 0
        if ctx.chain:
            W1b, W2b = W1c.to(torch.bfloat16), W2c.to(torch.bfloat16)
            h = torch.empty((M, F), dtype=torch.bfloat16, device=x.device)
            f = torch.empty((M, N), dtype=torch.float32, device=x.device)
            mbits = torch.empty((M, F // 32), dtype=torch.int32, device=x.device)      # "active and kept", one bit per hidden unit
            _chk(_lib().emloco_ffn_fwd(M, F, _p(x2), _p(W1b), _p(W2b), _p(b1.contiguous()), _p(b2.contiguous()), _p(h), _p(mbits), _p(f), float(drop_p),
                                       int(seed1) & 0xFFFFFFFF, int(seed2) & 0xFFFFFFFF, _st(x2)), "emloco_ffn_fwd")
            ctx.save_for_backward(x2, W1b, W2b, h, mbits)
            return f.view(*xs[:-1], N)
        # the hidden layer is the largest tensor of the step (M x 1024): bf16 in HBM in the reduced-precision mode
        # (the bf16-in-memory GEMM variants serve the 128-wide tiles and 8-byte-aligned rows only: small models keep fp32)
        h16 = _matmul_precision[0] == "bf16" and N > 32 and F > 32 and K > 32 and F % 4 == 0 and K % 4 == 0 and N % 4 == 0
        h = torch.empty((M, F), dtype=torch.bfloat16 if h16 else torch.float32, device=x.device)
        f32 = x2.dtype == torch.float32 and not h16
        gemm(1, M, F, K, x2, K, 0, 0, W1c, K, 0, 0, h, F, 0, bias=b1.contiguous(), flags=GEMM_BIAS | GEMM_RELU, drop_p=drop_p, drop_seed=seed1,
             b_image=_weight_image(W1c, M, F, K, K, 0) if f32 else None)
        f = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gemm(1, M, N, F, h, F, 0, 0, W2c, F, 0, 0, f, N, 0, bias=b2.contiguous(), flags=GEMM_BIAS, drop_p=drop_p, drop_seed=seed2,
             ksplit=_ksplit_for(F, M * N),                     # (whole for the predictor's tall batches; as LinearFn splits a small one)
             b_image=_weight_image(W2c, M, N, F, F, 0) if f32 else None)
        ctx.save_for_backward(x2, W1c, W2c, h)
        return f.view(*xs[:-1], N)

    @staticmethod
    def backward(ctx, df):
        dx, dW1, db1, dW2, db2 = _ffn_backward(ctx.saved_tensors[:5] if ctx.chain else tuple(ctx.saved_tensors[:4]) + (None,), ctx.chain, ctx.drop, ctx.xs,
                                               ctx.needs_input_grad[0], df)
        return dx, dW1, db1, dW2, db2, None, None, None


def _ffn_backward(saved, chain, drop, xs, need_dx, df):
    """The feed-forward block's backward pass (FeedForwardFn / FeedForwardNormFn): (dx, dW1, db1, dW2, db2) from the gradient w.r.t. its output."""
    if True:
        x2, W1, W2, h, mbits = saved
        M, K = x2.shape
        F, N = W1.shape[0], W2.shape[0]
        p, _, seed2 = drop
        lib, st, dev = _lib(), _st(x2), x2.device
        df2 = df.contiguous().view(M, N)
        db2 = torch.empty(N, dtype=torch.float32, device=dev)
        if p > 0.0:                                      # the output dropout's mask and linear2's bias gradient in one pass
            dz2 = torch.empty_like(df2)
            ws = torch.empty(lib.emloco_colsum_workspace(M, N), dtype=torch.float32, device=dev)
            _chk(lib.emloco_act_bwd_colsum(M, N, _p(df2), None, 0, p, seed2 & 0xFFFFFFFF, _p(dz2), _p(db2), _p(ws), st), "emloco_act_bwd_colsum")
        else:
            dz2, db2 = df2, colsum(df2)
        if chain:
            # (W1, W2 are the forward's bf16 copies: the weight-gradient products below never read them)
            dz1 = torch.empty((M, F), dtype=torch.bfloat16, device=dev)
            dx = torch.empty((M, K), dtype=torch.float32, device=dev)
            w2t, w1t = W2.t().contiguous(), W1.t().contiguous()       # (named: a temporary's block would be handed to the next allocation)
            # (round 6) the kernel leaves the column sums of dz1 per wave (M / 32 rows x F): linear1's bias gradient without a pass over dz1
            # (EMLOCO_FFN_COLSUM=0: the separate pass, A/B knob)
            cpart = torch.empty((lib.emloco_ffn_bwd_colsum_rows(M), F), dtype=torch.float32, device=dev) if _FFN_COLSUM else None
            _chk(lib.emloco_ffn_bwd_input_colsum(M, F, _p(dz2), _p(w2t), _p(w1t), _p(mbits), _p(dz1), _p(dx), float(p), _p(cpart), st), "emloco_ffn_bwd_input_colsum")
            dW2 = torch.empty((N, F), dtype=torch.float32, device=dev)
            gemm(1, N, F, M, dz2, N, 0, 1, h, F, 0, 1, dW2, F, 0, ksplit=_ksplit_for(M, N * F))          # dW2 = dz2^T h
            dW1 = torch.empty((F, K), dtype=torch.float32, device=dev)
            gemm(1, F, K, M, dz1, F, 0, 1, x2, K, 0, 1, dW1, K, 0, ksplit=_ksplit_for(M, F * K))         # dW1 = dz1^T x
            # (linear1's bias gradient is the column sum of dz1 as STORED -- bf16-rounded -- where the unchained bf16 path sums the fp32
            # values before rounding them: a difference of one bf16 rounding per element, inside the reduced-precision mode's 2e-2 bar;
            # summed inside the backward kernel per wave, folded here)
            return (dx.view(xs) if need_dx else None), dW1, colsum(cpart if cpart is not None else dz1), dW2, db2
        dW2 = torch.empty((N, F), dtype=torch.float32, device=dev)
        gemm(1, N, F, M, dz2, N, 0, 1, h, F, 0, 1, dW2, F, 0, ksplit=_ksplit_for(M, N * F), flags=_bwd_flags("dw"))          # dW2 = dz2^T h
        h16 = h.dtype == torch.bfloat16
        dz1 = torch.empty((M, F), dtype=h.dtype, device=dev)         # the hidden layer's gradient follows its dtype
        db1 = torch.empty(F, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.emloco_gemm_relu_bwd_workspace(M, F), dtype=torch.float32, device=dev)
        fl = (GEMM_BF16 | GEMM_C16 | GEMM_MASK16) if h16 else {"bf16": GEMM_BF16, "fp32_split": GEMM_SPLIT | _bwd_flags("dx")}.get(_matmul_precision[0], 0)
        # (a piece image holds three pieces: with the two-piece backward the matrix itself is the faster operand)
        img2 = _weight_image(W2, M, F, N, F, 1) if (not h16 and dz2.dtype == torch.float32 and _BWD_PIECES["dx"] == 3) else None     # B(n = hidden unit, k = output) = W2[k][n]
        _chk(lib.emloco_gemm_relu_bwd(M, F, N, _p(dz2), N, _p(img2 if img2 is not None else W2), F, 1, _p(dz1), _p(h), 1.0 / (1.0 - p), _p(db1), _p(ws),
                                      fl | (GEMM_B_SPLITIMG if img2 is not None else 0), st), "emloco_gemm_relu_bwd")
        dx = None
        if need_dx:
            dx = torch.empty((M, K), dtype=torch.float32, device=dev)
            gemm(1, M, K, F, dz1, F, 0, 0, W1, K, 0, 1, dx, K, 0, ksplit=1 if h16 else _ksplit_for(F, M * K),   # dx = dz1 W1
                 b_image=_weight_image(W1, M, K, F, K, 1) if (not h16 and _BWD_PIECES["dx"] == 3) else None, flags=_bwd_flags("dx"))
            dx = dx.view(xs)
        dW1 = torch.empty((F, K), dtype=torch.float32, device=dev)
        gemm(1, F, K, M, dz1, F, 0, 1, x2, K, 0, 1, dW1, K, 0, ksplit=_ksplit_for(M, F * K), flags=_bwd_flags("dw"))         # dW1 = dz1^T x
        return dx, dW1, db1, dW2, db2


_FFN_NORM = os.environ.get("EMLOCO_FFN_NORM", "1") != "0"      # the chained feed-forward also adds the residual and normalises (round 6)


class FeedForwardNormFn(torch.autograd.Function):
    """y = LayerNorm(res + dropout(linear2(dropout(relu(linear1(x)))))) gamma + beta -- the second half of the post-norm encoder layer
    (model_jta.py:177) as ONE forward launch in the reduced-precision mode (`emloco_ffn_fwd_norm`, round 6): the chained feed-forward
    kernel holds a row in a lane pair, so the residual add and the row's statistics cost one v_permlane32_swap; the block's own output is
    never written (a write and a read of M x 128 less than FeedForwardFn + LayerNormFn).  The backward is theirs: LayerNorm backward on
    the saved sum, then the feed-forward's.  x and res are two autograd edges (as `layer_norm(..., fork=True)` hands them out)."""

    @staticmethod
    def forward(ctx, x, res, W1, b1, W2, b2, gamma, beta, eps, drop_p, seed1, seed2, fork):
        xs = x.shape
        x2 = x.contiguous().view(-1, xs[-1])
        r2 = res.contiguous().view(-1, xs[-1])
        M, K = x2.shape
        F = W1.shape[0]
        dev = x.device
        W1b, W2b = W1.contiguous().to(torch.bfloat16), W2.contiguous().to(torch.bfloat16)
        h = torch.empty((M, F), dtype=torch.bfloat16, device=dev)
        mbits = torch.empty((M, F // 32), dtype=torch.int32, device=dev)
        y, xr = torch.empty((M, K), dtype=torch.float32, device=dev), torch.empty((M, K), dtype=torch.float32, device=dev)
        mean, rstd = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
        _chk(_lib().emloco_ffn_fwd_norm(M, F, _p(x2), _p(W1b), _p(W2b), _p(b1.contiguous()), _p(b2.contiguous()), _p(h), _p(mbits), _p(r2),
                                        _p(gamma), _p(beta), float(eps), _p(y), _p(xr), _p(mean), _p(rstd), float(drop_p),
                                        int(seed1) & 0xFFFFFFFF, int(seed2) & 0xFFFFFFFF, _st(x2)), "emloco_ffn_fwd_norm")
        ctx.save_for_backward(x2, W1b, W2b, h, mbits, xr, gamma, mean, rstd)
        ctx.xs, ctx.drop = xs, (float(drop_p), int(seed1), int(seed2))
        if fork:
            return y.view(xs), y.view(xs).detach()
        return y.view(xs)

    @staticmethod
    def backward(ctx, dy, dy_b=None):
        x2, W1b, W2b, h, mbits, xr, gamma, mean, rstd = ctx.saved_tensors
        rows, d = xr.shape
        if dy is None:
            dy, dy_b = dy_b, None
        dy2 = dy.contiguous().view(rows, d)
        dyb = dy_b.contiguous().view(rows, d) if dy_b is not None else None
        dxr = torch.empty_like(xr)
        dg = torch.empty(d, dtype=torch.float32, device=dy.device)
        db = torch.empty(d, dtype=torch.float32, device=dy.device)
        ws = torch.empty(_lib().emloco_layernorm_bwd_workspace(rows, d), dtype=torch.float32, device=dy.device)
        _chk(_lib().emloco_layernorm_bwd2(rows, d, _p(xr), _p(gamma), _p(mean), _p(rstd), _p(dy2), _p(dyb), _p(dxr), _p(dg), _p(db), _p(ws),
                                          _st(dy)), "emloco_layernorm_bwd")
        dx, dW1, db1, dW2, db2 = _ffn_backward((x2, W1b, W2b, h, mbits), True, ctx.drop, ctx.xs, ctx.needs_input_grad[0], dxr)
        return dx, dxr.view(ctx.xs), dW1, db1, dW2, db2, dg, db, None, None, None, None, None


def feed_forward_norm(x, res, W1, b1, W2, b2, gamma, beta, eps=1e-5, drop_p=0.0, fork=False):
    """layer_norm(feed_forward(x, ...), res, gamma, beta, eps, fork) -- as one forward launch where the chained kernel serves the shapes
    (reduced-precision mode, d = 128; EMLOCO_FFN_NORM=0: always the two ops)."""
    K = x.shape[-1]
    M = x.numel() // K
    if (_FFN_NORM and x.is_cuda and x.dtype == torch.float32 and res.dtype == torch.float32 and res.shape == x.shape and _ffn_chain_ok(M, K, W1.shape[0], W2.shape[0])
            and x.contiguous().data_ptr() % 16 == 0 and res.contiguous().data_ptr() % 16 == 0 and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0):
        s1, s2 = (next_dropout_seed(), next_dropout_seed()) if drop_p > 0.0 else (0, 0)
        if fork and not (torch.is_grad_enabled() and (x.requires_grad or res.requires_grad)):
            y = FeedForwardNormFn.apply(x, res, W1, b1, W2, b2, gamma, beta, float(eps), float(drop_p), s1, s2, False)
            return y, y
        return FeedForwardNormFn.apply(x, res, W1, b1, W2, b2, gamma, beta, float(eps), float(drop_p), s1, s2, bool(fork))
    return layer_norm(feed_forward(x, W1, b1, W2, b2, drop_p=drop_p), res, gamma, beta, eps, fork=fork)


def feed_forward(x, W1, b1, W2, b2, drop_p=0.0):
    """dropout(linear2(dropout(relu(linear1(x))))) with the two nn.Dropout(p) of the block (training: pass drop_p > 0)."""
    if drop_p > 0.0:
        s1 = next_dropout_seed()
        return FeedForwardFn.apply(x, W1, b1, W2, b2, float(drop_p), s1, next_dropout_seed())
    return FeedForwardFn.apply(x, W1, b1, W2, b2, 0.0, 0, 0)


class FusedAttentionFn(torch.autograd.Function):
    """Multi-head self-attention with head dim 32 in one kernel per direction (csrc/attention_kernels.hip): online
    softmax forward, probabilities recomputed from the saved log-sum-exp in the backward; the S x S scores never reach
    HBM.  qkv (Bn, S, 3 d) from the fused in-projection, key_pad (Bn, S) additive float bias."""
    MAX_SEQ_HEADS = 65535

    @staticmethod
    def forward(ctx, qkv, key_pad, nhead, drop_p=0.0, drop_seed=0, n_query=None):
        Bn, S, d3 = qkv.shape
        d = d3 // 3
        Sq = S if n_query is None else int(n_query)
        qkv = qkv.contiguous()
        key_pad = key_pad.contiguous() if key_pad is not None else None
        out = torch.empty((Bn, Sq, d), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((Bn * nhead, Sq), dtype=torch.float32, device=qkv.device)
        ctx.drop_p, ctx.drop_seed = float(drop_p), int(drop_seed)
        scale = 1.0 / float(d // nhead) ** 0.5
        lib, st = _lib(), _st(qkv)
        ctx.attn_flags = {"bf16": ATTN_BF16, "fp32_split": ATTN_SPLIT if _ATTN_SPLIT else 0}.get(_matmul_precision[0], 0)   # the backward follows the forward's choice
        if qkv.dtype == torch.bfloat16:
            ctx.attn_flags = ATTN_BF16 | ATTN_QKV16
        step = FusedAttentionFn.MAX_SEQ_HEADS // nhead
        for b0 in range(0, Bn, step):
            n = min(step, Bn - b0)
            _chk(lib.emloco_attention_fwd_queries(n, S, Sq, nhead, d, scale, _p(qkv, b0 * S * d3), _p(key_pad, b0 * S) if key_pad is not None else None,
                                                  _p(out, b0 * Sq * d), _p(lse, b0 * nhead * Sq), ctx.attn_flags, ctx.drop_p,
                                                  (ctx.drop_seed + b0) & 0xFFFFFFFF, st), "emloco_attention_fwd_queries")
        ctx.save_for_backward(qkv, key_pad, out, lse)
        ctx.nhead, ctx.scale, ctx.Sq = nhead, scale, Sq
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, key_pad, out, lse = ctx.saved_tensors
        nhead, scale, Sq = ctx.nhead, ctx.scale, ctx.Sq
        Bn, S, d3 = qkv.shape
        d = d3 // 3
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dsum = torch.empty_like(lse)
        lib, st = _lib(), _st(qkv)
        step = FusedAttentionFn.MAX_SEQ_HEADS // nhead
        for b0 in range(0, Bn, step):
            n = min(step, Bn - b0)
            _chk(lib.emloco_attention_bwd_queries(n, S, Sq, nhead, d, scale, _p(qkv, b0 * S * d3), _p(key_pad, b0 * S) if key_pad is not None else None,
                                                  _p(out, b0 * Sq * d), _p(lse, b0 * nhead * Sq), _p(dout, b0 * Sq * d), _p(dqkv, b0 * S * d3),
                                                  _p(dsum, b0 * nhead * Sq), ctx.attn_flags, ctx.drop_p, (ctx.drop_seed + b0) & 0xFFFFFFFF, st),
                 "emloco_attention_bwd_queries")
        return dqkv, None, None, None, None, None


def attention(qkv, key_pad, nhead, drop_p=0.0, n_query=None):
    """dropout(softmax(q k^T / sqrt(dh) + key_pad)) v per head (nn.MultiheadAttention; dropout on the probabilities in training
    mode: pass drop_p > 0): the fused kernels for head dim 32 (the shipped d = 128, 4 heads), the GEMM -> softmax -> GEMM
    composition otherwise.  n_query: only the first n_query rows of every sequence attend -> (Bn, n_query, d)."""
    if n_query is not None and n_query >= qkv.shape[1]:
        n_query = None
    if qkv.shape[-1] // 3 // nhead == 32:
        seed = next_dropout_seed() if drop_p > 0.0 else 0
        return FusedAttentionFn.apply(qkv, key_pad, nhead, float(drop_p), seed, n_query)
    out = AttentionFn.apply(qkv, key_pad, nhead, float(drop_p))
    return out if n_query is None else out[:, :n_query].contiguous()


class AttentionFn(torch.autograd.Function):
    """Multi-head self attention core on packed projections.

    qkv (Bn, S, 3*d) as produced by the in-projection (q | k | v), key_pad uint8 (Bn, S) or None -> (Bn, S, d).
    Per head: scores = q k^T (MFMA GEMM) -> masked softmax -> P v (MFMA GEMM); heads are addressed by pointer
    offset + leading dimension, so no permute copies are made.  P is kept for the backward (HBM is 288 GB)."""

    @staticmethod
    def forward(ctx, qkv, key_pad, nhead, drop_p=0.0):
        Bn, S, d3 = qkv.shape
        d = d3 // 3
        dh = d // nhead
        qkv = qkv.contiguous()
        P = torch.empty((nhead, Bn, S, S), dtype=torch.float32, device=qkv.device)
        # dropout on the probabilities (training): the mask multiplies the materialised P; the backward needs the dropped P for dV
        # and mask * dP before the softmax backward, which uses the undropped P
        M = None
        if drop_p > 0.0:
            M = (torch.rand((nhead, Bn, S, S), device=qkv.device) >= drop_p).float() / (1.0 - drop_p)
        out = torch.empty((Bn, S, d), dtype=torch.float32, device=qkv.device)
        scale = 1.0 / float(dh) ** 0.5
        lib = _lib()
        for h in range(nhead):
            Ph = P[h]
            gemm(Bn, S, S, dh, qkv, d3, S * d3, 0, qkv, d3, S * d3, 0, Ph, S, S * S, a_off=h * dh, b_off=d + h * dh)
            _chk(lib.emloco_softmax_fwd(Bn, S, S, scale, _p(Ph), _p(key_pad), _p(Ph), _st(qkv)), "emloco_softmax_fwd")
            Pd = Ph if M is None else (Ph * M[h]).contiguous()
            gemm(Bn, S, dh, S, Pd, S, S * S, 0, qkv, d3, S * d3, 1, out, d, S * d, b_off=2 * d + h * dh, c_off=h * dh)
        ctx.save_for_backward(qkv, P, M)
        ctx.nhead, ctx.scale = nhead, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P, M = ctx.saved_tensors
        nhead, scale = ctx.nhead, ctx.scale
        Bn, S, d3 = qkv.shape
        d = d3 // 3
        dh = d // nhead
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dP = torch.empty((Bn, S, S), dtype=torch.float32, device=qkv.device)
        lib = _lib()
        for h in range(nhead):
            Ph = P[h]
            # dP = dO v^T ; dV = P^T dO
            gemm(Bn, S, S, dh, dout, d, S * d, 0, qkv, d3, S * d3, 0, dP, S, S * S, a_off=h * dh, b_off=2 * d + h * dh)
            Pd = Ph if M is None else (Ph * M[h]).contiguous()
            gemm(Bn, S, dh, S, Pd, S, S * S, 1, dout, d, S * d, 1, dqkv, d3, S * d3, b_off=h * dh, c_off=2 * d + h * dh)
            if M is not None:
                dP.mul_(M[h])
            _chk(lib.emloco_softmax_bwd(Bn * S, S, scale, _p(Ph), _p(dP), _p(dP), _st(qkv)), "emloco_softmax_bwd")
            # dQ = dS k ; dK = dS^T q
            gemm(Bn, S, dh, S, dP, S, S * S, 0, qkv, d3, S * d3, 1, dqkv, d3, S * d3, b_off=d + h * dh, c_off=h * dh)
            gemm(Bn, S, dh, S, dP, S, S * S, 1, qkv, d3, S * d3, 1, dqkv, d3, S * d3, b_off=h * dh, c_off=d + h * dh)
        return dqkv, None, None, None


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) * gamma + beta (post-norm encoder layer)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, fork=False):
        # fork: the output is returned TWICE (y, y').  A post-norm layer's output feeds the next sublayer and its residual branch: with
        # one output autograd adds the two gradients in a pass of its own (read 2, write 1 over M x d) before this node's backward
        # reads the sum; with two outputs the backward gets both and adds them on load (emloco_layernorm_bwd2).
        shp = x.shape
        d = shp[-1]
        x2 = x.contiguous().view(-1, d)
        r2 = res.contiguous().view(-1, d) if res is not None else None
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        xr = torch.empty_like(x2) if r2 is not None else x2
        _chk(_lib().emloco_layernorm_fwd_save(rows, d, float(eps), _p(x2), _p(r2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd),
                                              _p(xr) if r2 is not None else None, _st(x)), "emloco_layernorm_fwd_save")
        ctx.save_for_backward(xr, gamma, mean, rstd)
        ctx.has_res, ctx.shp = res is not None, shp
        if fork:
            return y.view(shp), y.view(shp).detach()
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy, dy_b=None):
        xr, gamma, mean, rstd = ctx.saved_tensors
        rows, d = xr.shape
        if dy is None:
            dy, dy_b = dy_b, None
        dy2 = dy.contiguous().view(rows, d)
        dyb = dy_b.contiguous().view(rows, d) if dy_b is not None else None
        dxr = torch.empty_like(xr)
        dg = torch.empty(d, dtype=torch.float32, device=dy.device)
        db = torch.empty(d, dtype=torch.float32, device=dy.device)
        ws = torch.empty(_lib().emloco_layernorm_bwd_workspace(rows, d), dtype=torch.float32, device=dy.device)
        _chk(_lib().emloco_layernorm_bwd2(rows, d, _p(xr), _p(gamma), _p(mean), _p(rstd), _p(dy2), _p(dyb), _p(dxr), _p(dg), _p(db), _p(ws),
                                          _st(dy)), "emloco_layernorm_bwd")
        dxr = dxr.view(ctx.shp)
        return dxr, (dxr if ctx.has_res else None), dg, db, None, None


def layer_norm(x, res, gamma, beta, eps=1e-5, fork=False):
    """fork=True returns (y, y'): hand y to the next sublayer and y' to its residual branch (see LayerNormFn.forward)."""
    if fork and not (torch.is_grad_enabled() and (x.requires_grad or (res is not None and res.requires_grad))):
        y = LayerNormFn.apply(x, res, gamma, beta, eps)
        return y, y
    return LayerNormFn.apply(x, res, gamma, beta, eps, bool(fork))


class LocoValFn(torch.autograd.Function):
    """Fused LocoVal forward/backward; gradients flow to the six parameters and to the trajectory."""

    @staticmethod
    def forward(ctx, traj, pose, vel, w1, b1, w2, b2, w3, b3):
        B = traj.shape[0]
        ts = traj.shape[-1]
        traj_c, pose_c, vel_c = traj.contiguous().float(), pose.contiguous().float(), vel.contiguous().float()
        dev = traj.device
        value = torch.empty(B, device=dev)
        x100, h1, h2, ang = torch.empty(B, 100, device=dev), torch.empty(B, 49, device=dev), torch.empty(B, 24, device=dev), torch.empty(B, device=dev)
        ps = [t.contiguous() for t in (w1, b1, w2, b2, w3, b3)]
        _chk(_lib().emloco_locoval_fwd(B, _p(traj_c), ts, _p(pose_c), _p(vel_c), *[_p(t) for t in ps], _p(value), _p(x100), _p(h1), _p(h2),
                                       _p(ang), _st(traj)), "emloco_locoval_fwd")
        ctx.save_for_backward(traj_c, pose_c, vel_c, ps[0], ps[2], ps[4], value, x100, h1, h2, ang)
        ctx.mark_non_differentiable(x100)
        return value.view(B, 1), x100

    @staticmethod
    def backward(ctx, dvalue, _dx100):
        traj, pose, vel, w1, w2, w3, value, x100, h1, h2, ang = ctx.saved_tensors
        B, ts = traj.shape[0], traj.shape[-1]
        dev = traj.device
        dparams = torch.empty(6174, device=dev)
        dtraj = torch.empty_like(traj)
        ws = torch.empty(B * 6174, device=dev)
        dv = dvalue.contiguous().view(B).float()
        _chk(_lib().emloco_locoval_bwd(B, _p(traj), ts, _p(pose), _p(vel), _p(w1), _p(w2), _p(w3), _p(value), _p(x100), _p(h1), _p(h2), _p(ang),
                                       _p(dv), _p(dparams), _p(dtraj), _p(ws), _st(traj)), "emloco_locoval_bwd")
        o = [0, 4900, 4949, 6125, 6149, 6173, 6174]
        g = [dparams[o[i]:o[i + 1]] for i in range(6)]
        return (dtraj, None, None, g[0].view(49, 100), g[1], g[2].view(24, 49), g[3], g[4].view(1, 24), g[5])


def gemm_timing(enable=None):
    lib = _lib()
    if enable is not None:
        _chk(lib.emloco_gemm_enable_timing(int(enable)), "emloco_gemm_enable_timing")
        return None
    n, ms, fl = C.c_int(), C.c_float(), C.c_double()
    _chk(lib.emloco_gemm_timing_stats(C.byref(n), C.byref(ms), C.byref(fl)), "emloco_gemm_timing_stats")
    return n.value, ms.value, fl.value


def gemm_timing_bytes():
    """Algorithmic bytes (operands + output once, in their memory dtypes) of the launches the last `gemm_timing()` read summed."""
    b = C.c_double()
    _chk(_lib().emloco_gemm_timing_bytes(C.byref(b)), "emloco_gemm_timing_bytes")
    return b.value
