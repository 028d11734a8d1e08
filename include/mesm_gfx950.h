/*
 * mesm_gfx950.h — C-ABI of libmesm_gfx950.so, the MI355X (gfx950 / CDNA4) kernel
 * library behind the MESM training hot path (MESM.forward -> Criterion.forward ->
 * backward).
 *
 * The reference (lntzm/MESM) has no FFI boundary: its hot path is PyTorch ATen ops
 * called from model/model.py, model/transformer.py, model/attention.py,
 * model/position_encoding.py, model/criterion.py and model/matcher.py.  Each entry
 * point below replaces one ATen op *site class* of that path; the reference
 * file:line it stands in for is cited on the declaration.
 *
 * Conventions (all entries):
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless marked host;
 *   - stream-ordered: work is enqueued on `stream` (a hipStream_t passed as void*),
 *     nothing synchronises, nothing allocates; workspaces are passed in;
 *   - return 0 on success, a negative MESM_E* code on a rejected argument, and
 *     MESM_ELAUNCH - hipError on a failed launch; never throws;
 *   - fp32 everywhere ("f32" arithmetic; integer indices are int32/int64 as stated);
 *   - row-major tensors with explicit strides in ELEMENTS.
 */
#ifndef MESM_GFX950_H
#define MESM_GFX950_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MESM_OK 0
#define MESM_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define MESM_EALIGN (-2)   /* pointer or stride alignment not supported   */
#define MESM_ELAUNCH (-1000) /* launch failed: code = MESM_ELAUNCH - hipError_t */

/* ABI version; bumped whenever a signature changes. */
int mesm_abi_version(void);
/* Name of the compiled offload arch ("gfx950"). Host pointer, static storage. */
const char* mesm_arch(void);

/* ------------------------------------------------------------------------- */
/* Activation / transform selectors shared by the GEMM prologue and epilogue */
#define MESM_ACT_NONE 0
#define MESM_ACT_RELU 1
#define MESM_ACT_PRELU 2 /* single learnable slope, read from a device scalar */

/* Operand layout selectors: which index of the operand is contiguous. */
#define MESM_LAYOUT_REDUCE_CONTIG 0 /* element (i, r) at base + i*ld + r */
#define MESM_LAYOUT_OUTER_CONTIG 1  /* element (i, r) at base + r*ld + i */

/*
 * C[M,N] (+)= epilogue( opA(A)[M,K] @ opB(B)[K,N] ), exact-f32 MFMA
 * (v_mfma_f32_32x32x2_f32), LDS-tiled.
 *
 * Replaces: every nn.Linear / F.linear site on the path and their autograd
 * backward GEMMs — nn.MultiheadAttention in/out projections
 * (transformer.py:490,532,597,620,643), decoder sa_ / ca_ projections
 * (transformer.py:737-741,759-766,771,779), FFN linear1/linear2
 * (transformer.py:537,603,608,647,794), LinearLayer.net (model.py:421-431),
 * MLP heads (model.py:397-409, transformer.py:21-33), saliency projections
 * (model.py:301-302), MLM head (model.py:85-88).
 *
 * Operand indexing:  A is (i=m, r=k), B is (j=n, r=k); `*_layout` says which of the
 * two indices is contiguous, `ld*` is the stride of the other one.
 *
 * Prologue on A (and on B), applied while the operand is staged into LDS:
 *   x = A[m,k] (+ A2[m,k] if A2 != NULL, same strides)           [with_pos_embed]
 *   x = act(x)              a_act  in {NONE, RELU, PRELU(*slope)} [FFN activation]
 *   x = dropout(x)          a_drop_p > 0: keep iff hash(seed, idx) >= p; /(1-p), where idx is the
 *                           dense row-major index of the element in the operand AS STORED:
 *                           m*K + k for a reduce-contiguous A, k*M + m for an outer-contiguous
 *                           A (A = dY^T reads the mask that the forward epilogue wrote on Y)
 * B: same (B2 is the optional second addend); index k*N + n (outer-contiguous) or n*K + k.
 *
 * Epilogue, in this order (each step optional):
 *   t = acc * out_scale
 *   t += bias[n]
 *   pre_out[m,n] = t                 (optional second output, see the field)
 *   t = act(t)                       e_act in {NONE, RELU, PRELU(*slope)}
 *   t = dropout(t)                   e_drop_p > 0, index m*N + n
 *   t *= act'(aux[m,n])              e_actgrad in {NONE, RELU, PRELU}; for PRELU also
 *                                    *dslope += sum(t_before * min(aux,0)) (via dslope_ws)
 *   t += residual[m,n]
 *   C[m,n] = t | C[m,n] += t | atomicAdd(C[m,n], t)      (accumulate = 0|1|2)
 * split_k > 1 forces atomic accumulation (C must be initialised by the caller);
 * bias / residual are then added by the first split only, and act/dropout/actgrad
 * are rejected (MESM_EINVAL).
 *
 * colsum (optional): colsum[m] += sum_k opA(A)[m,k]  (atomic; used for bias grads,
 * where A = dY^T).
 */
typedef struct MesmGemmArgs {
  const float* A;
  const float* A2;
  const float* B;
  const float* B2;
  float* C;
  int32_t M, N, K;
  int32_t a_layout, b_layout;
  int64_t lda, ldb, ldc;
  const float* bias;
  const float* residual;
  int64_t ldr;
  const float* aux;
  int64_t ldaux;
  const float* slope;  /* device scalar for PRELU (prologue, epilogue, actgrad) */
  float* dslope;       /* device scalar accumulator (actgrad PRELU), may be NULL */
  float* colsum;       /* [M] accumulator, may be NULL */
  int32_t a_act, b_act;
  float a_drop_p, b_drop_p;
  uint32_t a_drop_seed, b_drop_seed;
  int32_t e_act, e_actgrad;
  float e_drop_p;
  uint32_t e_drop_seed;
  float out_scale;
  int32_t accumulate;
  int32_t split_k;
  /* optional device scalar added to every dropout seed of this launch at run time, so a
     captured HIP graph draws fresh masks on every replay (NULL = 0) */
  const uint32_t* seed_offset;
  /* workspace for the PRELU slope gradient: >= ceil(M/32) * ceil(N/32) * max(split_k, 1) floats,
     required when e_actgrad == PRELU and dslope != NULL (one plain store per workgroup, then a
     1-workgroup reduction adds the sum into *dslope) */
  float* dslope_ws;
  /* optional second output (NULL = off): pre_out[m,n] = acc * out_scale + bias, i.e. the value BEFORE
     e_act / e_drop, leading dimension ldpre.  The FFN's first GEMM writes z = x W1^T + b1 (kept for the
     backward's PReLU gradient) and a = dropout(prelu(z)) (the second GEMM's operand) in one pass
     (transformer.py:537, 603, 608, 647, 794).  Not with split_k / accumulate. */
  float* pre_out;
  int64_t ldpre;
  /* row offset added to the epilogue-dropout index: the launch computes rows [e_drop_row0, e_drop_row0 + M) of
     a taller output whose mask the backward replays over the whole tensor (a 4800-row FFN output issued as a
     4096-row launch that fills the 256 CUs in one round plus a remainder launch, mesm_amd/kernels.py) */
  int32_t e_drop_row0;
  int32_t reserved0;
} MesmGemmArgs;

int mesm_gemm_f32(const MesmGemmArgs* args, void* stream);

/*
 * n <= 64 INDEPENDENT GEMMs (no problem reads what another writes, except atomic accumulation
 * into the same gradient) issued together: the small ones (those mesm_gemm_f32 would hand to its
 * k-split kernel) share launches of up to 8 problems, whatever their shapes, layouts and fusions;
 * the others are launched one by one.  Replaces nothing new in the reference: it is how the
 * backward GEMM pairs (dW, dX) of every nn.Linear and the decoder's per-layer projections
 * (transformer.py:737-747, 759-784) are issued here -- a launch costs ~5 us whatever its size.
 */
int mesm_gemm_group(const MesmGemmArgs* args, int32_t n, void* stream);
/* The PReLU slope-gradient partials of a launch (dslope / dslope_ws) are reduced by the NEXT GEMM launch on the stream
 * (first workgroup, before its own work) instead of a launch of their own; dslope_ws must stay valid until then.
 * mesm_gemm_flush_side reduces whatever is still pending with plain launches: call it before dslope is read or
 * dslope_ws is released without another GEMM launch in between. */
int mesm_gemm_flush_side(void* stream);
/* Tuning tools only: the dispatch switches MESM_GEMM_TILE / MESM_GEMM_BF16X are read from the environment ONCE, when the
 * library is loaded; this entry changes them between calls of one process (a negative value keeps the current one). */
int mesm_gemm_set_switches(int32_t force_tile, int32_t bf16x);
/* The GEMM arithmetic in force: 2 = f32 products as THREE fp16 matrix products over operands split into two fp16 terms under a
 * wave-owned power-of-two scale (the default since round 6, csrc/gemm_ws.hpp), 6 = six bf16 matrix products over operands split
 * exactly into three bf16 terms (round 4), 0 = every product on the f32 matrix instruction (what bench.py reports as `dtype`). */
int mesm_gemm_get_bf16x(void);
/* Forget the pending reductions without running them (error paths: their workspaces may already be released). */
int mesm_gemm_drop_side(void);

/*
 * Launch-duration measurement of mesm_gemm_f32 (the dominant kernel of the step) for
 * bench.py's roofline object.  mesm_gemm_tape(1) starts recording the argument struct of every
 * GEMM launch (mesm_gemm_tape(0) stops); bench.py records while the step is captured into a HIP
 * graph, whose private memory pool keeps every recorded pointer valid afterwards.
 * mesm_gemm_tape_replay re-issues the recorded launches `reps` times back to back on `stream`,
 * every repetition bracketed by ONE pair of HIP events on that stream (a pair per launch costs
 * ~5 us of its own), synchronises (host-blocking; never inside a timed region) and returns the
 * summed elapsed time, the number of launches and their algorithmic FLOPs (2*M*N*K each):
 * average launch duration = total_ms / launches.
 */
int mesm_gemm_tape(int32_t record);
int mesm_gemm_tape_replay(void* stream, int32_t reps, double* total_ms, int64_t* launches,
                          double* total_flops, double* total_bytes);
/* algorithmic bytes of one pass over the recorded tape: A + B + C only, and with every side matrix of the fused
 * epilogues / prologues (second operands, residual, activation-gradient aux, accumulate's read of C, second output) */
int mesm_gemm_tape_bytes(double* operands, double* with_sides);
/* One recorded launch timed alone (`reps` back-to-back issues under one event pair) and what it carries:
 * shapes = up to 64 x (M, N, K, flags) int32, flags = split_k | a_layout << 8 | b_layout << 9 | grouped << 10.
 * Diagnostic (tools/tape_profile.py); mesm_gemm_tape_size = number of recorded launches. */
int mesm_gemm_tape_entry(void* stream, int32_t idx, int32_t reps, double* ms, int32_t* n_problems,
                         int32_t* shapes);
int mesm_gemm_tape_size(void);

/* ------------------------------------------------------------------------- */
/*
 * LayerNorm over the last dim, rows x D, eps inside the sqrt (torch semantics).
 * Replaces nn.LayerNorm sites: transformer.py:536,539,646,649,754,793,796,400,403;
 * model.py:430 (LinearLayer.LayerNorm, D = 2818/4098/512/300/256).
 * fwd writes y, and mean/rstd (rows) for the backward.
 * bwd writes dx (or dx += if accumulate) and atomically accumulates dgamma/dbeta;
 * dx == NULL (input needs no gradient) runs the parameter-gradient reduction only.
 * drop_p > 0 fuses the Dropout that follows the LayerNorm in LinearLayer (model.py:421-431):
 * fwd writes dropout(LN(x)) (keep iff hash(seed, row*D + col) >= p, scaled 1/(1-p)) and bwd
 * applies the same mask to dy while loading it.
 */
int mesm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int64_t rows, int32_t D, float eps,
                       float drop_p, uint32_t drop_seed, const uint32_t* seed_offset,
                       void* stream);
int mesm_layernorm_bwd(const float* dy, const float* x, const float* gamma,
                       const float* mean, const float* rstd, float* dx, float* dgamma,
                       float* dbeta, int64_t rows, int32_t D, int32_t accumulate_dx,
                       float drop_p, uint32_t drop_seed, const uint32_t* seed_offset,
                       void* stream);
/* Same, with a second output dx2 = dropout(dx; drop2_p, drop2_seed) (NULL = off): the mask of the block
 * whose output (+ residual) the LayerNorm normalised -- `norm(x + dropout(sublayer(x)))`,
 * transformer.py:534-539, 645-649, 753-754, 792-796 -- so that block's backward gets mask * dx without an
 * element-wise launch of its own.  Mask index = row * D + col (the sublayer output's dense index). */
int mesm_layernorm_bwd2(const float* dy, const float* x, const float* gamma,
                        const float* mean, const float* rstd, float* dx, float* dgamma,
                        float* dbeta, int64_t rows, int32_t D, int32_t accumulate_dx,
                        float drop_p, uint32_t drop_seed, const uint32_t* seed_offset,
                        float* dx2, float drop2_p, uint32_t drop2_seed, void* stream);

/* Forward with a second output y2 = y + add (add, y2: rows x D, both NULL = off): the `with_pos_embed` query
 * (transformer.py:512, 577, 640) of the attention block that consumes y, written by the LayerNorm that produces y
 * instead of an element-wise launch.  Backward counterpart: dyb (NULL = off) is the gradient of that second
 * consumer, added to dy on load; addend (NULL = off) is a gradient that reaches x on another route -- the residual
 * branch of `x + FFN(LN(x))` (transformer.py:536-538, 601-609) -- added to dx on store. */
int mesm_layernorm_fwd2(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                        float* rstd, int64_t rows, int32_t D, float eps, float drop_p, uint32_t drop_seed,
                        const uint32_t* seed_offset, const float* add, float* y2, void* stream);
int mesm_layernorm_bwd3(const float* dy, const float* x, const float* gamma, const float* mean,
                        const float* rstd, float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t D,
                        int32_t accumulate_dx, float drop_p, uint32_t drop_seed, const uint32_t* seed_offset,
                        float* dx2, float drop2_p, uint32_t drop2_seed, const float* dyb, const float* addend,
                        void* stream);

/*
 * Grouped launches: n independent LayerNorm problems (the same launch phase of independent chains of the step:
 * the enhance / SS-MESM / MLM stacks after the input projections, model.py:184-207, 307-332) in as few kernels
 * as possible -- problems with D <= 256 (one float4 per lane) share launches of up to 8, the others run through
 * the plain entry points.  One struct serves both directions: the forward reads x, gamma, beta, eps, drop_*, add
 * and writes y, mean, rstd, y2; the backward reads dy, x, gamma, mean, rstd, drop_*, drop2_*, dyb, addend and
 * writes dx, dx2, dgamma, dbeta (field meanings as in mesm_layernorm_fwd2 / mesm_layernorm_bwd3).
 */
typedef struct MesmLnArgs {
  const float* x;
  const float* gamma;
  const float* beta;
  float* y;
  float* mean;
  float* rstd;
  int64_t rows;
  int32_t D;
  float eps;
  float drop_p;
  uint32_t drop_seed;
  const uint32_t* seed_offset;
  const float* add;
  float* y2;
  /* backward */
  const float* dy;
  float* dx;
  float* dgamma;
  float* dbeta;
  int32_t accumulate_dx;
  float drop2_p;
  uint32_t drop2_seed;
  int32_t relu_in; /* backward, D <= 256 only: x = relu(z) came out of a Linear+ReLU; dx (and dx2) are written as d z,
                      i.e. masked by x > 0, so that block's own mask launch goes away (model.py:408,432) */
  float* dx2;
  const float* dyb;
  const float* addend;
} MesmLnArgs;
int mesm_layernorm_fwd_group(const MesmLnArgs* list, int32_t n, void* stream);
int mesm_layernorm_bwd_group(const MesmLnArgs* list, int32_t n, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Multi-head attention core: scores = scale * Q K^T -> mask(-inf) -> softmax ->
 * dropout -> P V, one (batch, head) per workgroup column, K/V tiles staged in LDS,
 * wavefront-shuffle row max / row sum, online softmax over key tiles.
 *
 * Replaces the attention core inside nn.MultiheadAttention at transformer.py:532,
 * 597,643 (torch/nn/functional.py multi_head_attention_forward) and the custom
 * multi_head_attention_forward core attention.py:329-386 (dk != dv allowed).
 *
 * Tensors: q (B, Lq, H*dk), k (B, Lk, H*dk), v (B, Lk, H*dv), o (B, Lq, H*dv) with
 * batch stride *_bs and row stride *_ls (elements); head h occupies columns
 * [h*dk, (h+1)*dk) of a row.  lse (B, H, Lq) = log-sum-exp of the masked scaled
 * scores (saved for the backward).
 *
 * Masks (uint8, nonzero = masked):
 *   kpad (B, Lk)   key_padding_mask;
 *   qpad (B, Lq)   only for mask_mode = MESM_MASK_T2V_QUIRK: reproduces
 *                  transformer.py:528-530 / :593-595, where the (B*H, Lq, Lk)
 *                  attn_mask built with .repeat(nhead,1,1) is indexed by
 *                  nn.MultiheadAttention as b*H+h:  key j of query i in (b,h) is
 *                  masked iff kpad[b,j] or (qpad[b2,i] and kpad[b2,j]),
 *                  b2 = (b*H + h) mod B.
 * A row whose keys are all masked yields NaN, like the reference.
 * Dropout on the probabilities: keep iff hash(seed, ((b*H+h)*Lq+i)*Lk+j) >= p.
 */
#define MESM_MASK_KPAD 0
#define MESM_MASK_T2V_QUIRK 1
/* kpad (optional) plus the causal rule of the CLIP text transformer (text_encoder.py:325-331, the additive
 * -inf upper triangle passed as attn_mask at :184): key j of query i is masked iff j > i.  Forward only. */
#define MESM_MASK_CAUSAL 2

typedef struct MesmAttnArgs {
  const float* q;
  const float* k;
  const float* v;
  float* o;
  float* lse;
  int32_t B, H, Lq, Lk, dk, dv;
  int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, o_bs, o_ls;
  const uint8_t* kpad;
  const uint8_t* qpad;
  int32_t mask_mode;
  float scale;
  float drop_p;
  uint32_t drop_seed;
  /* backward only */
  const float* d_o; /* (B, Lq, H*dv), strides o_bs/o_ls */
  float* dq;        /* strides q_bs/q_ls; MUST be zero-initialised when Lk > 64 */
  float* dk_;       /* strides k_bs/k_ls */
  float* dv_;       /* strides v_bs/v_ls */
  const uint32_t* seed_offset; /* see MesmGemmArgs.seed_offset */
  /* T2V_QUIRK with several independent batches stacked along B (positive and negative pass in
     one launch): the batch index wraps inside groups of mask_group rows,
     b2 = (b / G) * G + ((b % G) * H + h) mod G.  0 = one group of B rows. */
  int32_t mask_group;
  /* Split heads (NULL = off): head h's dk features are [ q[.., h*dk/2 : (h+1)*dk/2] || q2[.., same columns] ],
     i.e. q and q2 are (B, Lq, H*dk/2) tensors with the strides q_bs / q_ls (k, k2 likewise with k_bs / k_ls):
     the decoder cross-attention's per-head [content || position] queries and keys (transformer.py:778-784,
     attention.py embed_dim 2d) without materialising the interleaved (B, L, 2d) copies.  The backward writes
     the two halves of dq / dk into dq, dq2 / dk_, dk2 (same layouts). */
  const float* q2;
  const float* k2;
  float* dq2;
  float* dk2;
  /* with split heads: added to the FIRST half of every key head while it is staged (same layout / strides as k):
     decoder layer 0's key content = ca_kcontent_proj(memory) + ca_kpos_proj(pos) (transformer.py:773-776).  The
     backward's dk_ is then the gradient of that sum, i.e. of both terms. */
  const float* k_add;
  /* T2V_QUIRK only, NULL = off: the modulus of the batch wrap read from DEVICE memory at run time -- the number of valid
     pairs of a batch that was padded with dummy pairs up to a captured capacity; mask_group (or B) then is only the
     stride between stacked batches, and rows at or beyond the modulus see their own masks. */
  const int32_t* mask_mod;
} MesmAttnArgs;

int mesm_attn_fwd(const MesmAttnArgs* args, void* stream);
/* Needs q,k,v,o,lse from the forward plus d_o; writes dq,dk_,dv_. */
int mesm_attn_bwd(const MesmAttnArgs* args, void* stream);
/* Grouped launches of n independent attention problems (see mesm_layernorm_fwd_group): problems the matrix-core
 * forward (dk = dv = 32, Lk <= 128, no split heads) / the lane-per-key backward (dk = dv = 32, no split heads)
 * take share launches of up to 8, the others run through the plain entry points.  Same contracts as
 * mesm_attn_fwd / mesm_attn_bwd per problem. */
/* 1 if mesm_attn_bwd ADDS into dq / dq2 for this shape (they must then be zero on entry), 0 if it writes them. */
int mesm_attn_bwd_accumulates_dq(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t dk, int32_t dv, int32_t split);
int mesm_attn_fwd_group(const MesmAttnArgs* list, int32_t n, void* stream);
int mesm_attn_bwd_group(const MesmAttnArgs* list, int32_t n, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Sine position encoding of a validity mask.  Replaces
 * PositionEmbeddingSine.forward, position_encoding.py:51-72 (normalize=True,
 * scale=2*pi, temperature 10000): x = cumsum(mask); x = x/(x[-1]+1e-6)*2pi;
 * out[b,l,2i] = sin(x/T^(2i/D)), out[b,l,2i+1] = cos(x/T^(2i/D)).
 * mask (B, L) uint8 (nonzero = valid) -> out (B, L, D).
 */
int mesm_sine_pos_fwd(const uint8_t* mask, float* out, int32_t B, int32_t L, int32_t D,
                      void* stream);

/*
 * Query sine embedding of (center, width) reference points.  Replaces
 * gen_sineembed_for_position, transformer.py:43-59: ref (R, 2) -> out (R, D),
 * D/2 features for the centre then D/2 for the width, each interleaved sin/cos of
 * ref*2pi / 10000^(2*floor(i/2)/(D/2)).
 * bwd: dref (R, 2) (+)= sum_i dout * d(out)/d(ref).
 */
int mesm_query_sine_fwd(const float* ref, float* out, int64_t R, int32_t D, void* stream);
int mesm_query_sine_bwd(const float* ref, const float* dout, float* dref, int64_t R,
                        int32_t D, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Elementwise dropout with the library's counter hash (same stream of bits the GEMM
 * prologue/epilogue and the attention core use): y = x * keep/(1-p), keep iff
 * hash(seed, flat_index) >= p.  Replaces nn.Dropout sites not fused elsewhere and
 * lets tests materialise the exact mask.  In-place allowed (y == x).
 */
int mesm_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed,
                 const uint32_t* seed_offset, void* stream);

/*
 * Inference windows (eval.py:63-78): out (N, Q, 3) = [ (cx - w/2) * duration[n], (cx + w/2) * duration[n],
 * softmax(logits)[0] ] from logits (N, Q, 2), spans (N, Q, 2) = (centre, width), duration (N).
 */
int mesm_windows(const float* logits, const float* spans, const float* duration, float* out, int32_t N,
                 int32_t Q, void* stream);

/*
 * Decoder reference-point arithmetic, fused (each was ~8 / ~5 ATen launches, twice that in backward):
 *   ref_update   out = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid as transformer.py:36-40
 *                (clamp to [0,1], eps 1e-3): transformer.py:392-394 (new reference points) and
 *                model.py:250 (span head).  bwd: ddelta, dref (dref may be NULL).
 *   qsine_scale  out[r,:] = qsine[r,:] * scale[r,:] * sigmoid(anchor[r]) / ref[r,1]
 *                (transformer.py:370-376: query_scale modulation and the ref_anchor_head width
 *                modulation; scale may be NULL = layer 0).  bwd: dqsine, dscale, danchor (R), dref (R,2).
 */
int mesm_ref_update_fwd(const float* delta, const float* ref, float* out, int64_t n, float eps,
                        void* stream);
int mesm_ref_update_bwd(const float* out, const float* ref, const float* dout, float* ddelta,
                        float* dref, int64_t n, float eps, void* stream);
/* The decoder's first reference points (transformer.py:197, 361): out (N, QC) = sigmoid(p (QC)) for every pair;
 * backward dp[j] += out[0, j] (1 - out[0, j]) sum_n dout[n, j] into the parameter's gradient view (one workgroup). */
int mesm_ref_init_fwd(const float* p, float* out, int32_t N, int32_t QC, void* stream);
int mesm_ref_init_bwd(const float* out, const float* dout, float* dp, int32_t N, int32_t QC, void* stream);
/*
 * The reference points at a decoder layer boundary as ONE launch (transformer.py:343-397; round 6): the refined point,
 * its sine embedding (the next layer's ref_point_head input) and that embedding modulated by query_scale / ref_anchor_head
 * (the next layer's ca_qpos_sine_proj input) were ref_update | query_sine | qsine_scale, in front of layer 0
 * ref_init | query_sine.  R = pairs * queries rows, one wave per row, results bit-identical to the separate kernels.
 *   fwd   p != NULL (INIT): ref_out[r, c] = sigmoid(p[(2 r + c) mod QC]); else (NEXT): sigmoid(delta + inverse_sigmoid(prev)).
 *         qsine (R, D) = gen_sineembed_for_position(ref_out); anchor != NULL: qscaled (R, D) = qsine * scale *
 *         sigmoid(anchor[r]) / ref_out[r, 1] (scale may be NULL = 1).
 *   bwd   (NEXT; the reference detaches the refined point before embedding it, transformer.py:397) ddelta / dprev from
 *         dref_out (NULL = zero; dprev may be NULL), dscale / danchor from dqscaled (NULL = zero; dscale may be NULL);
 *         ddelta == NULL or danchor == NULL skips that half.
 *   init_sine_bwd   dp[j] += ref[j] (1 - ref[j]) sum_n (da + db + dc + query_sine_bwd(dqsine + dqsine2))[n, j]: the
 *         gradients of the initial points' consumers (stacked output, width modulation, first refinement) and of the
 *         embedding's (ref_point_head, modulation) are summed in the kernel; any may be NULL.  One wave per row; the
 *         pairs' terms meet by float atomics in dp.
 */
int mesm_ref_step_fwd(const float* p, int32_t QC, const float* delta, const float* prev, float eps,
                      const float* scale, const float* anchor, float* ref_out, float* qsine, float* qscaled,
                      int64_t R, int32_t D, void* stream);
int mesm_ref_step_bwd(const float* ref, const float* prev, const float* dref_out, float eps, const float* qsine,
                      const float* scale, const float* anchor, const float* dqscaled, float* ddelta, float* dprev,
                      float* dscale, float* danchor, int64_t R, int32_t D, void* stream);
int mesm_ref_init_sine_bwd(const float* ref, const float* da, const float* db, const float* dc, const float* dqsine,
                           const float* dqsine2, float* dp, int32_t N, int32_t QC, int32_t D, void* stream);
int mesm_qsine_scale_fwd(const float* qsine, const float* scale, const float* anchor,
                         const float* ref, float* out, int64_t R, int32_t D, void* stream);
int mesm_qsine_scale_bwd(const float* qsine, const float* scale, const float* anchor,
                         const float* ref, const float* dout, float* dqsine, float* dscale,
                         float* danchor, float* dref, int64_t R, int32_t D, void* stream);

/*
 * y = dropout(act(x)), flat index = element index (n % 4 == 0, 16-byte aligned): the FFN hidden
 * activation dropout(PReLU(linear1(.))) of transformer.py:537,603,608,647,794 written once.
 */
int mesm_act_dropout(const float* x, float* y, int64_t n, int32_t act, const float* slope, float p,
                     uint32_t seed, const uint32_t* seed_offset, void* stream);

/*
 * Activation backward + bias gradient: dz = dy * act'(ref), dbias[c] += sum_rows dz,
 * for RELU ref is the activation OUTPUT (y > 0), for PRELU ref is the
 * pre-activation (z) and *dslope += sum(dy * min(z,0)).  In-place allowed (dz == dy).
 * Replaces autograd of F.relu / nn.PReLU + bias at model.py:408,432-433 and
 * transformer.py:32,537,603,608,647,794.  act = NONE gives the plain bias gradient.
 */
int mesm_act_bias_bwd(const float* dy, const float* ref, float* dz, float* dbias,
                      const float* slope, float* dslope, int64_t rows, int32_t cols,
                      int32_t act, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Label-smoothed masked-LM NLL over the vocabulary (C classes) with row masks.
 * Replaces Criterion.cal_nll_loss, criterion.py:291-306 (eps = 0.1):
 *   logp = log_softmax(logit[r,:]);  nll_r = (1-eps)*(-logp[label_r]) + eps/C*(-sum logp)
 *   row_loss (R) = nll_r (0 where mask_r == 0);   correct (R) = argmax == label.
 * bwd: dlogit[r,c] = g_r * ( softmax - (1-eps)*onehot - eps/C ), g_r = per-row upstream
 * weight (already includes 1/mask.sum and 1/N from the reduction), 0 for masked rows.
 */
int mesm_nll_smooth_fwd(const float* logit, const int64_t* label, const uint8_t* mask,
                        float* row_loss, float* row_lse, uint8_t* correct, int64_t R,
                        int32_t C, float eps, void* stream);
int mesm_nll_smooth_bwd(const float* logit, const int64_t* label, const float* row_lse,
                        const float* row_grad, float* dlogit, int64_t R, int32_t C,
                        float eps, void* stream);

/*
 * Saliency losses of Criterion.loss_saliency, criterion.py:139-221, fused:
 * neg-pair BCE, the 11-stage rank-contrastive log-softmax over [pos || neg] scores
 * (tau 0.5, +1e-6 inside the log, -1e3 fill for padded clips), optional triplet hinge.
 * s_pos, s_neg (N, L) scores; label (N, L) float64; vmask (N, L) uint8 valid clips;
 * pos_idx / neg_idx (N, P) int64 (may be NULL -> no triplet term).
 * out_loss: device scalar (written).  bwd recomputes the row statistics and writes
 * ds_pos, ds_neg (N, L) = (*gscale) * dloss/ds  (gscale: device scalar, the upstream
 * gradient times the loss weight, so no host sync is needed).
 */
int mesm_saliency_loss_fwd(const float* s_pos, const float* s_neg, const double* label,
                           const uint8_t* vmask, const int64_t* pos_idx,
                           const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                           float rank_coef, float margin, float* out_loss, void* stream);
int mesm_saliency_loss_bwd(const float* s_pos, const float* s_neg, const double* label,
                           const uint8_t* vmask, const int64_t* pos_idx,
                           const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                           float rank_coef, float margin, const float* gscale,
                           float* ds_pos, float* ds_neg, void* stream);

/*
 * Hungarian matching cost + assignment on the device.  Replaces
 * HungarianMatcher.forward, matcher.py:39-117:
 *   C[b,q,t] = w_span * L1(span_cxw[b,q], tgt_cxw[t]) - w_giou * gIoU(xx(span), tgt_xx[t])
 *              - w_class * softmax(logits[b,q])[0]
 * and the optimal assignment between the T_b <= 64 targets of pair b and the Q <= 64 queries
 * (shortest-augmenting-path Hungarian in fp64 on the fp32 costs, the algorithm behind
 * scipy.optimize.linear_sum_assignment 1.9.1 pinned by the reference; identical result
 * whenever the optimum is unique).  Any shape within those extents, as the reference's call on
 * a (Q x T_b) block (matcher.py:108-117): with T_b < Q every target gets a query, with T_b >= Q
 * every query gets a target and T_b - Q targets stay unmatched.  The work arrays are fp64 in LDS
 * (60 KB per workgroup): MESM_EINVAL when one problem does not fit (never within the extents).
 * tgt_off (N+1) int32 prefix offsets into tgt_cxw / tgt_xx (sum T, 2).
 * cost (N, Q, Tmax) optional output; match_q (sum T) int32: query matched to each
 * target, in target order, -1 for a target left unmatched.
 */
int mesm_match(const float* logits, const float* spans, const float* tgt_cxw,
               const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
               int32_t Tmax, float w_span, float w_giou, float w_class, float* cost,
               int32_t* match_q, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused criterion blocks: one launch per loss and direction (criterion.hip).   */

/*
 * Hungarian match + span / gIoU / label losses of ONE decoder layer.  Replaces
 * HungarianMatcher.forward (matcher.py:39-117) followed by Criterion.loss_spans
 * (criterion.py:71-110) and Criterion.loss_labels (criterion.py:112-137):
 *   out4[0] = mean |src - tgt| over the matched (M x 2) elements, M = sum_b min(T_b, Q)
 *   out4[1] = mean (1 - gIoU(xx(src), tgt_xx)) over the matched pairs
 *   out4[2] = mean over N*Q of -log_softmax(logits)[cls] * {1, eos_coef}[cls]
 *             (cls = 0 on matched queries, 1 elsewhere; plain mean, quirk Q9)
 *   out4[3] = class_error = 100 - 100 * #(matched & argmax == 0) / M
 * match_q (sumT) int32 as in mesm_match.  One workgroup, deterministic sums.
 * bwd: dlogits, dspans (N, Q, 2) fully written; g3 = device pointer to the upstream
 * gradients of out4[0..2] (three consecutive floats).
 */
int mesm_set_loss_fwd(const float* logits, const float* spans, const float* tgt_cxw,
                      const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
                      int32_t Tmax, float w_span, float w_giou, float w_class, float eos_coef,
                      int32_t* match_q, float* out4, void* stream);
/* The same for n_layers <= 8 decoder layers (main + auxiliary outputs, criterion.py:338-357) in ONE launch, a workgroup per
 * layer: arrays of n_layers device pointers (host arrays); targets, weights and n_valid (may be NULL) shared. */
int mesm_set_loss_fwd_layers(const float* const* logits, const float* const* spans, int32_t n_layers,
                             const float* tgt_cxw, const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
                             int32_t Tmax, float w_span, float w_giou, float w_class, float eos_coef,
                             int32_t* const* match_q, float* const* out4, const int32_t* n_valid, void* stream);
int mesm_set_loss_bwd(const float* logits, const float* spans, const float* tgt_cxw,
                      const float* tgt_xx, const int32_t* tgt_off, const int32_t* match_q,
                      int32_t N, int32_t Q, float eos_coef, const float* g3, float* dlogits,
                      float* dspans, void* stream);

/*
 * SS-MESM reconstruction loss, Criterion.loss_rec_ss (criterion.py:223-274, "ablation 3"):
 * masked mean of projed_video_feat pv (N, Lv, D) over the GT clips cmask and of
 * expanded_words_feat ew (N, Le, D) over wmask, F.normalize (eps 1e-12), sim = cn wn^T / tau,
 * logits = sim - rowmax, log_prob = logits - log(sum exp + 1e-6),
 * loss = mean_n -(sum_k pos[n,k] log_prob[n,k]) / (sum_k pos[n,k] + 1e-6).
 * pos (N, N) uint8 is target-only (block-diagonal gIoU >= gamma) and built by the caller.
 * Saved for backward: cn, wn (N, D), stats (2N, 4): rows < N = {#clips, #words, |clip|, |words|},
 * rows >= N scratch for the per-row loss terms,
 * sim (N, N).  bwd writes dpv (N, Lv, D) and dew (N, Le, D) in full; g = device scalar.
 */
int mesm_rec_ss_fwd(const float* pv, const uint8_t* cmask, int32_t Lv, const float* ew,
                    const uint8_t* wmask, int32_t Le, const uint8_t* pos, int32_t N, int32_t D,
                    float tau, float* cn, float* wn, float* stats, float* sim, float* out,
                    void* stream);
int mesm_rec_ss_bwd(const float* cn, const float* wn, const uint8_t* pos, const float* sim,
                    const float* stats, const uint8_t* cmask, const uint8_t* wmask, int32_t N,
                    int32_t D, int32_t Lv, int32_t Le, float tau, const float* g, float* dpv,
                    float* dew, void* stream);

/*
 * Reductions around mesm_nll_smooth_* for Criterion.loss_rec_fw (criterion.py:276-304):
 * out2[0] = mean_n( sum_w row_loss[n,w] / #valid words of n ), out2[1] = masked accuracy.
 * rowgrad: row_grad[n,w] = (*g) * mask[n,w] / (N * #valid words of n)  -> mesm_nll_smooth_bwd.
 */
int mesm_rec_fw_reduce(const float* row_loss, const uint8_t* correct, const uint8_t* mask,
                       int32_t N, int32_t Lw, float* out2, void* stream);
int mesm_rec_fw_rowgrad(const uint8_t* mask, int32_t N, int32_t Lw, const float* g,
                        float* row_grad, void* stream);

/*
 * Saliency score (model.py:301-302): s[n,l] = <a[n,l,:], b[n,:]> * scale with
 * a = saliency_proj1(memory) (N, L, D), b = saliency_proj2(memory_global) (N, D),
 * scale = 1/sqrt(hidden_dim).  bwd: da = ds (x) b * scale, db = sum_l ds * a * scale.
 */
int mesm_rowdot_fwd(const float* a, const float* b, int32_t N, int32_t L, int32_t D, float scale,
                    float* s, void* stream);
int mesm_rowdot_bwd(const float* a, const float* b, const float* ds, int32_t N, int32_t L,
                    int32_t D, float scale, float* da, float* db, void* stream);

/*
 * MESM.post_process_text (model.py:145-152): words = F.normalize(x, eps 1e-5) when
 * `normalize`, wmask = (sum_c words != 0), sent = normalize(sum_w words / #valid).
 * x, words (N, Lw, D); wmask (N, Lw) uint8; sent (N, D).  No gradient (inputs are features).
 */
int mesm_text_prep(const float* x, int32_t N, int32_t Lw, int32_t D, int32_t normalize,
                   float* words, uint8_t* wmask, float* sent, void* stream);

/*
 * total = sum_k weights[k] * vals[k] (Criterion.forward, criterion.py:361-365; entries with
 * weight 0 are skipped so logged-only values may be non-finite) and its backward
 * out[k] = (*g) * weights[k].  All device pointers.
 */
int mesm_weighted_sum(const float* vals, const float* weights, int32_t n, float* out,
                      void* stream);
int mesm_scale_vec(const float* g, const float* weights, int32_t n, float* out, void* stream);

/*
 * The criterion forward (criterion.py:319-367: every loss of the step and their weighted total) in three launches:
 * A = the first stage of every block as workgroup ranges of one grid (set losses with their matching, one workgroup per
 * decoder layer | saliency | rec_ss masked means | masked-LM NLL rows), B = rec_ss' similarity rows, C = one workgroup
 * that finishes rec_fw and rec_ss and writes total = sum_k weights[k] * lv[k].  lv = the loss vector (n_slots floats;
 * the `*_slot` fields say where a block's values go: set losses 4 slots [span, giou, label, class_error], rec_fw 2
 * [loss, accuracy], the others 1).  Outputs kept for the backward: set_match, row_lse (row_loss / correct: staging),
 * cn / wn / stats (2N x 4) / sim.  Field meanings as in mesm_set_loss_fwd_layers / mesm_saliency_loss_fwd_nv /
 * mesm_nll_smooth_fwd + mesm_rec_fw_reduce_nv / mesm_rec_ss_fwd_nv / mesm_weighted_sum; results bit-identical to those.
 */
typedef struct MesmCritFwd {
  const float* weights;
  float* lv;
  float* total;
  const int32_t* n_valid; /* device scalar: real pairs of a padded batch, or NULL */
  int32_t N, n_slots;
  int32_t n_set, Q, Tmax; /* set-loss layers (<= 8), moment queries, most targets of a pair */
  float w_span, w_giou, w_class, eos_coef;
  int32_t reserved0;
  const float* tgt_cxw;
  const float* tgt_xx;
  const int32_t* tgt_off;
  const float* set_logits[8];
  const float* set_spans[8];
  int32_t* set_match[8];
  int32_t set_slot[8];
  int32_t sal_on, sal_L, sal_P, sal_slot;
  float rank_coef, margin;
  const float* s_pos;
  const float* s_neg;
  const double* sal_label;
  const uint8_t* vmask;
  const int64_t* pos_idx;
  const int64_t* neg_idx;
  int32_t fw_on, fw_Lw, fw_C, fw_slot;
  float fw_eps;
  int32_t reserved1;
  const float* logit;
  const int64_t* label;
  const uint8_t* words_mask;
  float* row_loss;
  float* row_lse;
  uint8_t* correct;
  int32_t ss_on, ss_D, ss_Lv, ss_Le, ss_slot;
  float ss_tau;
  const float* pv;
  const uint8_t* cmask;
  const float* ew;
  const uint8_t* wmask;
  const uint8_t* ss_pos;
  float* cn;
  float* wn;
  float* stats;
  float* sim;
} MesmCritFwd;
int mesm_criterion_fwd(const MesmCritFwd* args, void* stream);

/*
 * The whole criterion backward (criterion.py:319-367, the autograd of every loss of the step) as ONE launch: the
 * gradient kernels of the set losses (one per decoder layer), the saliency loss, the masked-LM NLL and rec_ss are
 * independent 256-thread kernels and run as workgroup ranges of one grid.  g_total = d total (1 float), weights = the loss
 * vector's weights (the `*_slot` fields index it: a role multiplies g_total by its own weight, mesm_scale_vec is folded in);
 * the NLL's per-row weights (mesm_rec_fw_rowgrad) are computed in place from words_mask.  A block with its `*_on` flag 0 (or
 * n_set = 0) is skipped.  Field meanings as in mesm_set_loss_bwd_nv / mesm_saliency_loss_bwd_nv / mesm_nll_smooth_bwd /
 * mesm_rec_ss_bwd_nv.
 */
typedef struct MesmCritBwd {
  const float* g_total;
  const float* weights;
  const int32_t* n_valid; /* device scalar: real pairs of a padded batch, or NULL */
  int32_t N, Q;           /* pairs, moment queries */
  int32_t n_set;          /* set-loss layers (<= 8) */
  float eos_coef;
  const float* tgt_cxw;
  const float* tgt_xx;
  const int32_t* tgt_off;
  const float* set_logits[8];
  const float* set_spans[8];
  const int32_t* set_match[8];
  float* set_dlogits[8];
  float* set_dspans[8];
  int32_t set_slot[8];
  int32_t sal_on, sal_L, sal_P, sal_slot;
  float rank_coef, margin;
  const float* s_pos;
  const float* s_neg;
  const double* sal_label;
  const uint8_t* vmask;
  const int64_t* pos_idx;
  const int64_t* neg_idx;
  float* ds_pos;
  float* ds_neg;
  int32_t fw_on, fw_Lw, fw_C, fw_slot;
  float fw_eps;
  int32_t reserved0;
  const float* logit;
  const int64_t* label;
  const float* row_lse;
  const uint8_t* words_mask;
  float* dlogit;
  int32_t ss_on, ss_D, ss_Lv, ss_Le, ss_slot;
  float ss_tau;
  const float* cn;
  const float* wn;
  const uint8_t* ss_pos;
  const float* sim;
  const float* stats;
  const uint8_t* cmask;
  const uint8_t* wmask;
  float* dpv;
  float* dew;
} MesmCritBwd;
int mesm_criterion_bwd(const MesmCritBwd* args, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Assembly kernels of MESM.forward (csrc/glue.hip): the concatenations, repeats, selections and masked
 * replacements between the transformer blocks, one launch each, with the matching gradient joins.
 *
 * mesm_stack_rows      n <= 8 tensors at once: dst_t (2N rows) = [ src_t ; src_t[idx] ] (gather[t] = 1) or
 *                      [ src_t ; src_t ] (0); row_bytes[t] bytes per row, any dtype.  The positive and the
 *                      negative pass stacked along the batch: model.py:260-299 with neg_index from
 *                      sample_outclass_neg (data_utils.py:113-124).  src / dst / row_bytes / gather: HOST arrays.
 * mesm_unstack_rows    its backward for one float tensor (R floats per row, R % 4 == 0):
 *                      dx[i] = d2[i] + sum_{j: idx[j] == i} d2[N + j]   (idx NULL: j == i)
 * mesm_prepend_fwd     xo (B, L+1, D) = [tok ; x]; optional po = [ptok ; pos], xp = xo + po and the key padding
 *                      mask pado = [first_pad ; pad].  tok_per_row: tok is (B, D) (the reconstructed sentence token in
 *                      front of the words, model.py:221-224) instead of (D) (global token + its position,
 *                      transformer.py:185-188, model.py:236-238).
 * mesm_prepend_bwd     g = dxo + dxp: dx = g[:, 1:]; dtok = g[:, 0] (per row: stored; shared: atomically ADDED over
 *                      b into the gradient view); dptok += (dxp + dpo)[:, 0].  NULL inputs count as zero.
 * mesm_split_token_*   mem (B, L+1, D) -> g = mem[:, 0], loc = mem[:, 1:], dec = loc[:Bd] (transformer.py:196-198; the
 *                      decoder sees the positive half only); backward dmem = assembled sum, missing parts zero.
 * mesm_token_mix_*     y[r] = m2[r] ? tok2 : (m1[r] ? tok1 : x[r])  (model.py:361-394 _replace_unknown / _mask_words,
 *                      :493-501 masked sentence slot); backward dx = dy on unmasked rows else 0, dtok1 / dtok2 += the
 *                      column sums of dy over their rows (atomic, into the gradient views).
 * mesm_gather_rows_*   y[j] = valid[j] ? x[idx[j]] : 0, optionally L2-normalised like F.normalize (eps 1e-12; rnorm[j]
 *                      keeps the norm).  Backward through the host-built inverse map inv (source row -> j or -1):
 *                      every source row is written once (zeros where not gathered): model.py:312-325, :485-486.
 * mesm_add_wrap        out[i] = a[i] + b[i mod nb]  (n, nb element counts, % 4 == 0)
 * mesm_skinny_linear_bwd  backward of y = x W^T + b with J <= 4 output features (last layers of the MLP heads
 *                      span_embed / bbox_embed / ref_anchor_head and class_embed: model.py:118-119, 402-410,
 *                      transformer.py:349-352) in one launch: dx (M, K) = dz (M, J) W (J, K), zeroed where x <= 0 when
 *                      relu_mask (x is the previous layer's ReLU output); dw (J, K) += dz^T x and db (J) += colsum(dz),
 *                      both atomic into the gradient views.  dx / db may be NULL.
 */
int mesm_stack_rows(const void* const* src, void* const* dst, const int64_t* row_bytes, const int32_t* gather,
                    int32_t n, const int64_t* idx, int32_t N, void* stream);
int mesm_unstack_rows(const float* d2, const int64_t* idx, float* dx, int32_t N, int64_t R, void* stream);
int mesm_prepend_fwd(const float* tok, const float* x, const float* ptok, const float* pos, const uint8_t* pad,
                     float* xo, float* po, float* xp, uint8_t* pado, int32_t B, int32_t L, int32_t D,
                     int32_t tok_per_row, int32_t first_pad, void* stream);
int mesm_prepend_bwd(const float* dxo, const float* dxp, const float* dpo, float* dx, float* dtok, float* dptok,
                     int32_t B, int32_t L, int32_t D, int32_t tok_per_row, void* stream);
int mesm_split_token_fwd(const float* mem, float* g, float* loc, float* dec, int32_t B, int32_t L, int32_t D,
                         int32_t Bd, void* stream);
int mesm_split_token_bwd(const float* dg, const float* dloc, const float* ddec, float* dmem, int32_t B, int32_t L,
                         int32_t D, int32_t Bd, void* stream);
int mesm_token_mix_fwd(const float* x, const uint8_t* m1, const float* tok1, const uint8_t* m2, const float* tok2,
                       float* y, int64_t rows, int32_t D, void* stream);
int mesm_token_mix_bwd(const float* dy, const uint8_t* m1, const uint8_t* m2, float* dx, float* dtok1, float* dtok2,
                       int64_t rows, int32_t D, void* stream);
int mesm_gather_rows_fwd(const float* x, const int64_t* idx, const uint8_t* valid, float* y, float* rnorm,
                         int64_t rows, int32_t D, int32_t normalize, void* stream);
int mesm_gather_rows_bwd(const float* dy, const float* y, const float* rnorm, const int64_t* inv,
                         const uint8_t* valid, float* dx, int64_t src_rows, int32_t D, int32_t normalize,
                         void* stream);
int mesm_add_wrap(const float* a, const float* b, float* out, int64_t n, int64_t nb, void* stream);
/* out[i] = srcs[0][i] + ... + srcs[k - 1][i], 1 <= k <= 8, n % 4 == 0, 16-byte aligned: the gradient of a tensor with
 * several consumers in one launch (mesm_amd.ops.fork) instead of the autograd engine's k - 1 pairwise adds. */
int mesm_add_n(const float* const* srcs, int32_t k, float* out, int64_t n, void* stream);

/*
 * Assembly problems of one launch phase in ONE launch (mesm_glue_group): independent members -- none reads what another
 * writes -- of the kinds below, with the argument meaning of the plain entry point of the same name.  op-specific
 * fields: p = pointers in the order of that entry point's pointer arguments, n / i = its extents:
 *   TOKEN_MIX_FWD   p: x, m1, tok1, m2, tok2, y          n[0] rows         i[0] D
 *   TOKEN_MIX_BWD   p: dy, m1, m2, dx, dtok1, dtok2      n[0] rows         i[0] D
 *   GATHER_ROWS_FWD p: x, idx, valid, y, rnorm           n[0] rows         i[0] D, i[1] normalize
 *   GATHER_ROWS_BWD p: dy, y, rnorm, inv, valid, dx      n[0] source rows  i[0] D, i[1] normalize
 *   UNSTACK_ROWS    p: d2, idx, dx                       n[0] R            i[0] N
 *   STACK_ROWS      p: src, dst, idx (NULL: repeat)      n[0] row bytes    i[0] N      (one tensor of mesm_stack_rows)
 *   ADD_TILE        p: a, b, out   out[k] = a[k % na] + b[k % nb]: n[0] = elements of out, n[1] = na, i[0..1] = nb as int64
 *   GATHER_ADD      p: a, b, idx, valid, y       y[j] = valid[j] ? a[idx[j]] + b[idx[j]] : 0: n[0] rows, i[0] D
 */
enum {
  MESM_GLUE_TOKEN_MIX_FWD = 1,
  MESM_GLUE_TOKEN_MIX_BWD = 2,
  MESM_GLUE_GATHER_ROWS_FWD = 3,
  MESM_GLUE_GATHER_ROWS_BWD = 4,
  MESM_GLUE_UNSTACK_ROWS = 5,
  MESM_GLUE_STACK_ROWS = 6,
  MESM_GLUE_ADD_TILE = 7,
  MESM_GLUE_GATHER_ADD = 8
};
typedef struct MesmGlueArgs {
  int32_t op;
  int32_t reserved0;
  int32_t i[6];
  int64_t n[2];
  const void* p[8];
} MesmGlueArgs;
int mesm_glue_group(const MesmGlueArgs* list, int32_t n, void* stream);

/* n <= 32 byte ranges (16-byte aligned, multiples of 16 bytes) set to zero in one launch: the zero-initialised scratch of a
 * step (replaces one torch fill per step; no reference counterpart: the reference's tensors are fresh allocations). */
int mesm_fill_ranges(void* const* ptrs, const int64_t* nbytes, int32_t n, void* stream);
/* First node of a captured training step (no counterpart in the reference: its forward draws on the host and indexes
 * with host tensors, model.py:260, 361-384): copies slot (*pull_ctr % slots) of a ring of `slots` x slot_bytes in PINNED
 * HOST memory (device-readable) to dst, then *pull_ctr += 1 and, when given, *seed_ctr += 1 (the dropout seed offset of
 * the step).  slot_bytes % 16 == 0; one workgroup. */
int mesm_step_begin(const void* host_ring, int32_t slot_bytes, int32_t slots, void* dst, int32_t* pull_ctr,
                    int32_t* seed_ctr, void* stream);
int mesm_skinny_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db,
                           int64_t M, int32_t K, int32_t J, int32_t relu_mask, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Frozen text encoders (forward only, the reference runs them under no_grad).
 * fp16 tensors travel as void* (IEEE binary16, row-major).
 *
 * mesm_clip_embed      x[r, :] = fp16(tok[ids[r], :]) + fp16(pos[r % L, :])           (rows = N*L, fp16 out)
 *                      CLIPTextEncoder.forward, text_encoder.py:342-344 (fp32 tables, .type(fp16))
 * mesm_layernorm_f16   y = fp16(LayerNorm_fp32(float(x)))                              text_encoder.py:154-160
 * mesm_gemm_f16        C[M,N] = epi(A[M,K] @ W[N,K]^T + bias): fp16 operands, fp32 accumulation on
 *                      v_mfma_f32_32x32x8_f16, ONE rounding to fp16 after the bias, then optionally
 *                      QuickGELU (x * sigmoid(1.702 x), each fp16 op rounded, :163-165) and the residual add
 *                      (x = x + ..., :187-188).  A may be fp32 (a_is_f32: rounded to fp16 while staged -- the
 *                      attention output); C may be written as fp32 (c_is_f32: the fp16-rounded values widened,
 *                      input of mesm_attn_fwd).  K % 32 == 0; lda, ldw % 8 == 0.  Replaces the nn.Linear sites
 *                      of ResidualAttentionBlock (:172-177) and nn.MultiheadAttention's in/out projections.
 * mesm_text_pool       MESM.CLIP_encode_text / GloVe_encode_text tail (model.py:118-134, 138-143): first Lw
 *                      tokens of x (N, Lx, D) (fp16 or fp32), pads zeroed by mask (N, Lm) uint8, sentence =
 *                      masked mean of the un-normalised words, both L2-normalised (eps 1e-5) if normalize.
 * mesm_embed_rows      out[r, :] = table[ids[r], :]   GloveTextEncoder.forward (text_encoder.py:446-454)
 */
int mesm_clip_embed(const int64_t* ids, const float* tok, const float* pos, void* x, int64_t rows, int32_t L,
                    int32_t D, int32_t vocab, void* stream);
int mesm_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int32_t D,
                       float eps, void* stream);
int mesm_gemm_f16(const void* A, int32_t a_is_f32, int64_t lda, const void* W, int64_t ldw, const void* bias,
                  const void* residual, int64_t ldr, void* C, int32_t c_is_f32, int64_t ldc, int32_t M,
                  int32_t N, int32_t K, int32_t quick_gelu, void* stream);
int mesm_text_pool(const void* x, int32_t x_is_f16, const uint8_t* mask, int32_t N, int32_t Lx, int32_t Lm,
                   int32_t Lw, int32_t D, int32_t normalize, float* words, float* sent, void* stream);
int mesm_embed_rows(const int64_t* ids, const float* table, float* out, int64_t rows, int32_t D, int32_t vocab,
                    void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Optimizer tail on the flat buffers.  Replaces, for all trainable tensors at
 * once, nn.utils.clip_grad_norm_(model.parameters(), grad_clip) + torch.optim.AdamW.step()
 * (train.py:70-72, runner.py:348-352; amsgrad off, eps 1e-8):
 *   mesm_grad_sumsq   partial sums of squares of g (n % 4 == 0), one per workgroup; *np_out (host)
 *                     = number of partials (<= 1024); *step (device int32, may be NULL) += 1
 *   mesm_clip_grad    g *= min(1, max_norm / (||g|| + 1e-6));  *norm_out = ||g|| (optional)
 *   mesm_adamw_step   with coef = that clip factor when max_norm > 0 (1 otherwise), t = *step,
 *                     lr = *lr (device scalars: graph-capturable, StepLR writes lr between steps):
 *                       p *= 1 - lr*wd;  m = b1 m + (1-b1) coef g;  v = b2 v + (1-b2) (coef g)^2
 *                       p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 *                     active4 (n/4 bytes, optional): groups of 4 elements with 0 are skipped (tensors
 *                     whose gradient is None this step: no decay, no state update).
 */
int mesm_grad_sumsq(const float* g, int64_t n, float* partials, int32_t* np_out, int32_t* step,
                    void* stream);
int mesm_clip_grad(float* g, int64_t n, const float* partials, int32_t np, float max_norm,
                   float* norm_out, void* stream);
int mesm_adamw_step(float* p, const float* g, float* m, float* v, const uint8_t* active4, int64_t n,
                    const float* partials, int32_t np, float max_norm, const float* lr, float beta1,
                    float beta2, float eps, float weight_decay, const int32_t* step, float* norm_out,
                    void* stream);

/* ------------------------------------------------------------------------- */
/*
 * The loss blocks with a VALID-PAIR COUNT read from device memory (`n_valid`, NULL = N): the pairs [*n_valid, N)
 * are padding that brought a batch up to the capacity a HIP graph was captured with (the reference's loaders emit a
 * different number of pairs almost every batch, dataset/base.py:116-162).  Padding pairs take no part in the matching,
 * in any sum or in any denominator (every mean is over *n_valid), and their gradient rows are written as zeros.
 * Otherwise identical to the entry points of the same name without the suffix.
 */
int mesm_set_loss_fwd_nv(const float* logits, const float* spans, const float* tgt_cxw, const float* tgt_xx,
                         const int32_t* tgt_off, int32_t N, int32_t Q, int32_t Tmax, float w_span, float w_giou,
                         float w_class, float eos_coef, int32_t* match_q, float* out4, const int32_t* n_valid,
                         void* stream);
int mesm_set_loss_bwd_nv(const float* logits, const float* spans, const float* tgt_cxw, const float* tgt_xx,
                         const int32_t* tgt_off, const int32_t* match_q, int32_t N, int32_t Q, float eos_coef,
                         const float* g4, float* dlogits, float* dspans, const int32_t* n_valid, void* stream);
int mesm_saliency_loss_fwd_nv(const float* s_pos, const float* s_neg, const double* label, const uint8_t* vmask,
                              const int64_t* pos_idx, const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                              float rank_coef, float margin, float* out_loss, const int32_t* n_valid, void* stream);
int mesm_saliency_loss_bwd_nv(const float* s_pos, const float* s_neg, const double* label, const uint8_t* vmask,
                              const int64_t* pos_idx, const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                              float rank_coef, float margin, const float* gscale, float* ds_pos, float* ds_neg,
                              const int32_t* n_valid, void* stream);
int mesm_rec_ss_fwd_nv(const float* pv, const uint8_t* cmask, int32_t Lv, const float* ew, const uint8_t* wmask,
                       int32_t Le, const uint8_t* pos, int32_t N, int32_t D, float tau, float* cn, float* wn,
                       float* stats, float* sim, float* out, const int32_t* n_valid, void* stream);
int mesm_rec_ss_bwd_nv(const float* cn, const float* wn, const uint8_t* pos, const float* sim, const float* stats,
                       const uint8_t* cmask, const uint8_t* wmask, int32_t N, int32_t D, int32_t Lv, int32_t Le,
                       float tau, const float* g, float* dpv, float* dew, const int32_t* n_valid, void* stream);
int mesm_rec_fw_reduce_nv(const float* row_loss, const uint8_t* correct, const uint8_t* mask, int32_t N, int32_t Lw,
                          float* out2, const int32_t* n_valid, void* stream);
int mesm_rec_fw_rowgrad_nv(const uint8_t* mask, int32_t N, int32_t Lw, const float* g, float* row_grad,
                           const int32_t* n_valid, void* stream);

/* ------------------------------------------------------------------------- */
/*
 * Data-parallel gradient exchange over RCCL, owned by this library (new functionality: the reference is
 * single-process; insertion point train.py:68-72, between loss.backward() and clip_grad_norm_).  One process per
 * GPU.  Rank 0 draws the 128-byte unique id (mesm_ddp_unique_id) and hands it to the other ranks through any side
 * channel (mesm_amd.ddp uses the torch.distributed store); every rank then calls mesm_ddp_init.
 * mesm_ddp_allreduce sums buf[0, count) (fp32, in place) over the ranks: side = 0 on `stream` itself, side = 1 on the
 * communicator's own stream behind everything enqueued on `stream` so far (overlap with the rest of backward);
 * mesm_ddp_wait makes `stream` wait for the collectives issued with side = 1.  All three record into a HIP graph when
 * `stream` is capturing.  RCCL is bound at run time (dlopen of librccl.so.1); mesm_ddp_last_error() names a failure.
 */
int mesm_ddp_unique_id(uint8_t* out128);
int mesm_ddp_init(const uint8_t* id128, int32_t rank, int32_t world, void** handle);
int mesm_ddp_allreduce(void* handle, float* buf, int64_t count, void* stream, int32_t side);
int mesm_ddp_wait(void* handle, void* stream);
int mesm_ddp_destroy(void* handle);
/* ranks the communicator spans as RCCL reports it (ncclCommCount) */
int mesm_ddp_count(void* handle, int32_t* ranks);
const char* mesm_ddp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MESM_GFX950_H */
