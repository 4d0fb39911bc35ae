// policy_args.h -- argument blocks of the policy kernels (policy_kernels.hip), shared with the C ABI file (mms_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mms {

constexpr int kMaxGroups = 32;      // networks per grouped launch (MMS_MAX_GROUPS in include/mms.h)

// y_g = act(x_g w_g^T + b_g) for g < groups: one launch for all of them (same M, N, K)
struct LinearArgs {
    const float* x[kMaxGroups];
    const float* w[kMaxGroups];
    const float* b[kMaxGroups];
    float* y[kMaxGroups];
    int M, N, K;
    int act;        // 0: identity, 1: ELU (alpha = 1), 2: ReLU, 3: tanh
    // LayerNorm folded into the layers on either side of it (grouped MARL inference; all NULL otherwise).
    //   part_out[g] != NULL: the epilogue also leaves, per output row, the sum and the sum of squares of its 64 activations:
    //     part_out[g][(slot * M + row) * 2 + {0, 1}], slot = 2 * column_tile + wave half, N / 64 slots -- no atomics, so the
    //     statistics (launch_row_stats sums the slots in order) do not depend on block scheduling.
    //   stat_in[g] != NULL: x_g is the PRE-LayerNorm activation h and the layer is W (LN(h) gamma + beta) + b, evaluated as
    //     rstd (W~ h - mean s) + c with W~ = W diag(gamma) (passed as w_g), s = W~ 1 (s_g), c = W beta + b (passed as b_g) and
    //     (mean, rstd) per row from stat_in[g][row * 2 + {0, 1}].
    const float* s[kMaxGroups];
    const float* stat_in[kMaxGroups];
    float* part_out[kMaxGroups];
};

// (mean, 1 / sqrt(var + eps)) per row from the slot partials a layer's epilogue left (LinearArgs::part_out): stat_g[row * 2 + {0, 1}]
struct RowStatsArgs {
    const float* part[kMaxGroups];
    float* stat[kMaxGroups];
    int64_t M;
    int slots, width;       // width = the layer's N (the number of activations per row)
    float eps;
};

// y_g[r, 0:K] = LayerNorm(x_g[r, 0:K]) * gamma_g + beta_g, y_g[r, K:Kp] = 0 (row pitch Kp >= K): nn.LayerNorm over the last
// dimension, biased variance, eps inside the square root.  x_g == y_g with Kp == K is the in-place form.
struct LayerNormArgs {
    const float* x[kMaxGroups];
    const float* gamma[kMaxGroups];
    const float* beta[kMaxGroups];
    float* y[kMaxGroups];
    int64_t M;
    int K, Kp;
    int x_pitch;    // floats between consecutive input rows (>= K): K for a dense matrix, A * K for one agent's rows of an [N, A, K] block
    float eps;
    int stats_only; // 1: y_g[r * 2 + {0, 1}] = (mean, 1 / sqrt(var + eps)) of row r instead of the normalised row (gamma / beta unused)
};

// The output layer of each network on LayerNorm(h_g): out_g[r, j] = b_g[j] + sum_k w_g[j, k] LN(h_g[r])[k], j < A_g <= 16.
// std_g != NULL: a diagonal Gaussian is sampled around it (action = out + std z, z from the counter-based stream keyed by
// (seed + g, row_offset + r, counters_g[r]++, j)) and the per-dimension log-densities go to logp_g [M, A_g]; std_g == NULL: out_g is stored as is.
struct HeadsArgs {
    const float* h[kMaxGroups];
    const float* gamma[kMaxGroups];
    const float* beta[kMaxGroups];
    const float* w[kMaxGroups];
    const float* b[kMaxGroups];
    const float* std[kMaxGroups];
    float* out[kMaxGroups];
    float* logp[kMaxGroups];
    int64_t* counters[kMaxGroups];
    int A[kMaxGroups];
    int out_pitch[kMaxGroups];      // floats between consecutive rows of out_g (and logp_g): A_g dense, or the row width of an [N, agents, A] block
    uint64_t seed;
    int64_t M, row_offset;
    int H;
    float eps;
};

// The same layer with both operands in the three-plane bf16 format "P32" (split_kernels.hip): x_g [M, KC, 3, 32] bf16, w_g [N, KC, 3, 32]
// bf16, b_g [N] f32; y_g either P32 planes [M, N / 32, 3, 32] (out_planes, the next split layer's input) or f32 [M, N].
struct SplitLinearArgs {
    const void* x[kMaxGroups];
    const void* w[kMaxGroups];
    const float* b[kMaxGroups];
    void* y[kMaxGroups];
    int M, N, KC;       // KC = ceil(K / 32) chunks per row
    int act;            // as LinearArgs::act
    int out_mode;       // 0: y_g f32 [M, N]; 1: y_g P32 planes; 2: no y_g, only head_part (needs the LayerNorm folds)
    // LayerNorm folds (all NULL: none; given: all three, act = ELU) -- the meaning of LinearArgs::s / stat_in, and
    //   part_out[g][(slot * M + row) * 2 + {0, 1}] = (sum, sum of squared deviations from the slot's own mean) of the row's 64
    //   activations in slot = column / 64 (launch_row_stats_chan combines the slots; two-pass form: no cancellation)
    const float* s[kMaxGroups];
    const float* stat_in[kMaxGroups];
    float* part_out[kMaxGroups];
    // out_mode 2: head_w[g] [hdims[g], N] f32 (the output head's weight, the last LayerNorm's gamma folded in);
    //   head_part[g][(slot * M + row) * stride_g + j] = sum over the slot's 64 columns of y[row, n] head_w[j, n], j < hdims[g] <= 16,
    //   stride_g = hdims[g] rounded up to 4 (a critic's one value: 16 bytes per row and slot, an 8-action actor's: 32)
    const float* head_w[kMaxGroups];
    float* head_part[kMaxGroups];
    int hdims[kMaxGroups];
    int tiles;          // (set by the launcher) output tiles of the launch: the persistent grid walks them
};

// The finish of an output head whose dot products a split layer left as per-slot partials (SplitLinearArgs::head_part), with the last
// LayerNorm folded: dot_j = sum over slots of head_part, (mean, rstd) of the row from the slots' (sum, M2) pairs (part), and
// out[r, j] = rstd (dot_j - mean hs[j]) + hc[j]   (hs = head_w 1, hc = w beta + b), then the Gaussian sample exactly as HeadsArgs.
struct HeadsFinishArgs {
    const float* part[kMaxGroups];          // [slots, M, 2]
    const float* head_part[kMaxGroups];     // [slots, M, A_g rounded up to 4]
    const float* hs[kMaxGroups];            // [A_g]
    const float* hc[kMaxGroups];            // [A_g]
    const float* std[kMaxGroups];
    float* out[kMaxGroups];
    float* logp[kMaxGroups];
    int64_t* counters[kMaxGroups];
    int A[kMaxGroups];
    int out_pitch[kMaxGroups];
    uint64_t seed;
    int64_t M, row_offset;
    int slots, width;                       // width = slots * 64 = the hidden size
    float eps;
};

// P32 planes of `groups` matrices of the same shape in one launch (x_g rows at pitch x_pitch floats)
struct SplitPlanesArgs {
    const float* x[kMaxGroups];
    void* planes[kMaxGroups];
    int64_t rows;
    int K, x_pitch;
};

// The same layer with both operands as TWO scaled fp16 planes per fp32 number, format "H32" (split16_kernels.hip): x_g [M, KC, 2, 32] f16
// holding hi = f16(x xs[row]), lo = f16((x xs[row] - hi) 2^11) with xs[row] a power of two per row; w_g likewise with ws[row].
//   xinv[g][m] = 1 / xs (f32 [M]), winv[g][n] = 1 / ws (f32 [N]): the epilogue's exact rescaling of the accumulators
//   yscale[g][m] (out_mode 1): the power of two the output row m is multiplied with before it is split into the y planes
struct Split16LinearArgs {
    const void* x[kMaxGroups];
    const void* w[kMaxGroups];
    const float* b[kMaxGroups];
    void* y[kMaxGroups];
    const float* xinv[kMaxGroups];
    const float* winv[kMaxGroups];
    const float* yscale[kMaxGroups];
    int M, N, KC;
    int act, out_mode;              // as SplitLinearArgs
    const float* s[kMaxGroups];     // LayerNorm folds, as SplitLinearArgs
    const float* stat_in[kMaxGroups];
    float* part_out[kMaxGroups];
    const float* head_w[kMaxGroups];
    float* head_part[kMaxGroups];
    int hdims[kMaxGroups];
    int tiles, grid;                // (set by the launcher)
    uint64_t* clock_probe;          // optional (mms_layer_clock_probe): block 0 stores its life in shader cycles and in 100-MHz ticks
};

// H32 planes of `groups` matrices of the same shape (x_g rows at pitch x_pitch floats) with a power-of-two scale per row taken from the
// row's largest magnitude (scale[g][row], inv[g][row] = 1 / scale), and -- nchains > 0 -- the scales of the layers this row then
// flows through: chain[g] = [nchains][L][2] (mult, add) with bound_{l+1} = mult_l bound_l + add_l starting from the row's largest
// magnitude (mult = the layer's largest weight-row 1-norm, add = its largest |bias|; mult = 0 for a layer behind a LayerNorm, whose
// output bound does not depend on its input), chain_scale[g] / chain_inv[g] = [nchains][L][rows].
struct Split16PlanesArgs {
    const float* x[kMaxGroups];
    void* planes[kMaxGroups];
    float* scale[kMaxGroups];
    float* inv[kMaxGroups];
    const float* chain[kMaxGroups];
    float* chain_scale[kMaxGroups];
    float* chain_inv[kMaxGroups];
    float* stat[kMaxGroups];                // optional: (mean, 1 / sqrt(var + eps)) per row, f32 [rows, 2]
    float* l1[kMaxGroups];                  // optional: the row's 1-norm sum_k |x[r, k]|, f32 [rows] (a weight matrix: its layer's entry of the bound chain)
    int64_t rows;
    int K, x_pitch;
    int nchains, L;
    float eps;
    // per_group != 0: matrices of DIFFERENT shapes in one launch (the weights of all hidden layers at a refresh): group g is
    // [rows_g[g], K_g[g]], dense (pitch K_g[g]); rows / K / x_pitch above are then unused and the chain / stat outputs are not available
    int per_group;
    int64_t rows_g[kMaxGroups];
    int K_g[kMaxGroups];
};

// The bound chain refreshed on the device (no atomics, no host synchronisation: graph-capturable): entry e = c * L + l of the chain is
// (max_i l1[e][i], max_i |bias[e][i]|), i < n[e] -- layer l of chain c's (mult, add) pair from the row 1-norms its weight split left
// (Split16PlanesArgs::l1) and its bias; written to chain[e * 2 + {0, 1}] by block 0.  rows > 0: every block also evaluates the chain's
// scales for rows whose input bound is the constant bound0 (chain_scale / chain_inv [nchains, L, rows]).
struct ChainRefreshArgs {
    const float* l1[kMaxGroups];
    const float* bias[kMaxGroups];
    int n[kMaxGroups];
    float* chain;
    int nchains, L;
    float bound0;
    int64_t rows;
    float* chain_scale;
    float* chain_inv;
};

// The weights' side of a layer behind a folded LayerNorm (fold16_kernels.hip): per network g and row n of w_g [rows_g, K_g]
//   W~[n, k] = w[n, k] gamma[k] (gamma NULL: w itself), planes_g / inv_g = its H32 planes and inverse row scales (planes NULL: none),
//   wt_g = W~ as f32 (NULL: not stored), s_g[n] = sum_k W~[n, k], c_g[n] = sum_k w[n, k] beta[k] + bias[n] (beta / bias NULL: without),
//   rb_g[n] = |W~ row|_2 sqrt(K) + |c[n]|: the row's bound of W~ xhat + c over normalised inputs (|xhat|_2 <= sqrt(K)).
struct FoldPlanesArgs {
    const float* w[kMaxGroups];
    const float* gamma[kMaxGroups];
    const float* beta[kMaxGroups];
    const float* bias[kMaxGroups];
    void* planes[kMaxGroups];
    float* inv[kMaxGroups];
    float* s[kMaxGroups];
    float* c[kMaxGroups];
    float* rb[kMaxGroups];
    float* wt[kMaxGroups];
    int64_t rows_g[kMaxGroups];
    int K_g[kMaxGroups];
};

// Per network g: scale = 2^(14 - e), 1.001 max_{i < n[g]} rb_g[i] <= 2^e -> scale1[g][0] (optional) and ysc[g][0..M) / yinv[g][0..M) = 1 / scale
struct FoldScalesArgs {
    const float* rb[kMaxGroups];
    float* scale1[kMaxGroups];
    float* ysc[kMaxGroups];
    float* yinv[kMaxGroups];
    int n[kMaxGroups];
    int64_t M;
};

hipError_t launch_fold_planes16(const FoldPlanesArgs& a, int groups, hipStream_t s);
hipError_t launch_fold_scales16(const FoldScalesArgs& a, int groups, hipStream_t s);
hipError_t launch_split16_planes_group(const Split16PlanesArgs& a, int groups, hipStream_t s);
hipError_t launch_chain_refresh16(const ChainRefreshArgs& a, hipStream_t s);
hipError_t launch_linear_split16(const Split16LinearArgs& a, int groups, hipStream_t s);
void set_split16_clock_probe(uint64_t* out, int slots);   // mms_layer_clock_probe
hipError_t launch_linear_act(const LinearArgs& a, int groups, hipStream_t s);
hipError_t launch_row_stats_chan(const RowStatsArgs& a, int groups, hipStream_t s);
hipError_t launch_marl_heads_finish(const HeadsFinishArgs& a, int groups, hipStream_t s);
hipError_t launch_split_planes_group(const SplitPlanesArgs& a, int groups, hipStream_t s);
hipError_t launch_linear_split(const SplitLinearArgs& a, int groups, hipStream_t s);
hipError_t launch_split_planes(const float* x, void* planes, int64_t rows, int K, int x_pitch, hipStream_t s);
hipError_t launch_row_stats(const RowStatsArgs& a, int groups, hipStream_t s);
hipError_t launch_layernorm(const LayerNormArgs& a, int groups, hipStream_t s);
hipError_t launch_marl_heads(const HeadsArgs& a, int groups, hipStream_t s);

}  // namespace mms
