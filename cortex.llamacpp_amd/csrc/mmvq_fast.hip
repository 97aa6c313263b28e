// mmvq_fast.hip — single-token quantised mat-vec for the decode step, K % 2048 == 0 (every Llama-class shape).
//
// Same arithmetic as mmvq.hip (integer partial sums identical to ggml_vec_dot_q{4,5,6}_K_q8_K),
// restructured after measurement on MI355X (tools/bench_stream.hip, tools/bench_mmvq.hip):
//   * the 16-byte-per-lane load pattern alone streams at 5.3-5.7 TB/s; what slowed the first kernels down was the
//     serial  load -> wait -> decode/dot  chain inside each wave.  Here every wave is persistent and software-
//     pipelined: the loads of unit n+1 (2 rows x 2 passes = 8 x 16 B per lane) are in flight while unit n is decoded,
//     with two statically named register sets;
//   * the decode is branch-free (selects instead of divergent branches), uses 24-bit multiplies, and the Q8_K
//     activation slice of a pass is read once and applied to both rows of the pair;
//   * for K <= 4096 the lane's activation slices (2 passes) live in registers for the whole kernel, so the inner loop
//     touches neither LDS nor L2 for activations.
#include "mmvq_fast_dev.h"

namespace mi355 {

namespace {

// NT threads per workgroup, ONE workgroup per CU (the activation is staged / normalised / quantised once per CU)
template <int KB, int NT, int FUSE>
__global__ __launch_bounds__(NT) void mmvq_fast_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NW = NT / 64;
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
    int bl = (int)blockIdx.x - a.seg_block0[s];
    int sel_j = 0;
    if (a.n_sel > 1) {                       // the selected experts of one token share the launch: an even share of workgroups each
        const int per = nblk / a.n_sel;
        sel_j = bl / per < a.n_sel ? bl / per : a.n_sel - 1;
        bl -= sel_j * per;
        nblk = sel_j == a.n_sel - 1 ? nblk - sel_j * per : per;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = bl * NW + wave;
    const int nw = nblk * NW;
    switch (a.seg[s].type) {
        case T_Q4_K: run_fast<T_Q4_K, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j); break;
        case T_Q5_K: run_fast<T_Q5_K, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j); break;
        case T_Q6_K: run_fast<T_Q6_K, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j); break;
        case T_Q2_K: run_fast<T_Q2_K, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j); break;
        case T_Q3_K: run_fast<T_Q3_K, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j); break;
        case T_Q8_0:                                           // pre-quantised Q8_0 planes only in the register form (K <= 4096)
            if constexpr (FUSE != 0 || KB <= 2) run_fast<T_Q8_0, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j);
            break;
        case T_Q4_0:                                           // the other Q8_0-activation formats: same staging, same limits
            if constexpr (FUSE != 0 || KB <= 2) run_fast<T_Q4_0, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j);
            break;
        case T_Q5_0:
            if constexpr (FUSE != 0 || KB <= 2) run_fast<T_Q5_0, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j);
            break;
        case T_IQ4_NL:
            if constexpr (FUSE != 0 || KB <= 2) run_fast<T_IQ4_NL, KB, NT, FUSE>(a, a.seg[s], smem, gw, nw, NoSync(), sel_j);
            break;
        default: break;
    }
}

}  // namespace

bool mmvq_fast_applicable(const MMVQArgs &a) {
    if (a.T != 1 || (a.K % 256) != 0 || a.K <= 0) return false;
    const int kb = (a.K + 2047) >> 11;                         // passes of 8 super-blocks; the last one may be partial
    if (!mmvq_fast_kb_ok(kb)) return false;
    const int n = a.epi == EPI_SWIGLU ? 2 : a.n_seg;
    for (int s = 0; s < n; s++) {
        const int t = a.seg[s].type;
        if (t != T_Q4_K && t != T_Q5_K && t != T_Q6_K && t != T_Q2_K && t != T_Q3_K && !act_is_q80(t)) return false;
        // segments may mix K-quants and Q8_0 (8-expert files keep attn_k / attn_v in Q8_0): every workgroup stages the
        // activation in the format of ITS segment; pre-quantised planes must exist in that format
        if (a.fuse_mode == 0) {
            if (act_is_q80(t) && (kb > 2 || !a.aq0 || !a.ad0)) return false;
            if (!act_is_q80(t) && (!a.aq || !a.ad || !a.abs)) return false;
        }
    }
    if (a.fuse_mode < 0 || a.fuse_mode > 2) return false;
    if (a.fuse_mode == 1 && a.K > 8192) return false;          // RMSNorm + quantise: the hidden size; quantise-only (2): any listed K
    if (a.n_sel > 1 && (a.n_sel > 8 || (a.epi == EPI_SWIGLU ? 1 : a.n_seg) != 1 || a.epi == EPI_ADD)) return false;   // experts of one token: one tensor (or one gate/up pair)
    return true;
}

static int g_fast_nt = 0;     // 0 = pick per launch; tools/bench_mmvq.hip forces 512 / 768 to compare
void mmvq_fast_set_threads(int nt) { g_fast_nt = (nt == 512 || nt == 768) ? nt : 0; }

size_t mmvq_fast_plan(MMVQArgs &a, int max_blocks, int nwv) {
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return 0;
    const int n_work_seg = a.epi == EPI_SWIGLU ? 1 : a.n_seg;
    const int kb = (a.K + 2047) >> 11;
    size_t bytes[3] = {0, 0, 0}, total = 0;
    int want[3] = {0, 0, 0}, sum_want = 0;
    for (int s = 0; s < n_work_seg; s++) {
        bytes[s] = (size_t)a.seg[s].n_rows * a.seg[s].row_bytes;
        total += bytes[s];
        const int pairs = a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
        want[s] = (pairs + nwv - 1) / nwv * (a.n_sel > 1 ? a.n_sel : 1);
        sum_want += want[s];
    }
    a.seg_block0[0] = 0;
    for (int s = 0; s < n_work_seg; s++) {
        int nb = want[s];
        if (sum_want > max_blocks) {
            nb = (int)((double)max_blocks * (double)bytes[s] / (double)total + 0.5);
            if (nb < 1) nb = 1;
            if (nb > want[s]) nb = want[s];
        }
        a.seg_block0[s + 1] = a.seg_block0[s] + nb;
    }
    if (a.seg_block0[n_work_seg] > max_blocks) {               // rounding pushed the total over: trim the largest range
        int big = 0;
        for (int s = 1; s < n_work_seg; s++) if (a.seg_block0[s + 1] - a.seg_block0[s] > a.seg_block0[big + 1] - a.seg_block0[big]) big = s;
        const int over = a.seg_block0[n_work_seg] - max_blocks;
        for (int s = big; s < n_work_seg; s++) a.seg_block0[s + 1] -= over;
    }
    for (int s = n_work_seg; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n_work_seg];
    if (a.epi == EPI_SWIGLU) a.n_seg = 1;
    const size_t K = (size_t)kb * 2048;                        // whole passes
    size_t lds = K + (((K >> 8) * 4 + 15) & ~(size_t)15) + (((K >> 4) * 2 + 15) & ~(size_t)15);
    lds = (lds + 15) & ~(size_t)15;
    a.red_off = (int)lds;
    lds += 128;
    return lds;
}

hipError_t launch_mmvq_fast(MMVQArgs a, hipStream_t st) {
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return hipErrorInvalidValue;
    const int kb = (a.K + 2047) >> 11;
    // persistent grid, one workgroup per CU.  8 waves per CU measured better than 12 on the whole decode step
    // (tools/profile_decode.py: 2.05 vs 2.13 ms / token); MI355_MMVQ_NT=768 selects the 12-wave form for K <= 4096
    int nt = 512;
    if (kb <= 2 && g_fast_nt == 768) nt = 768;
    const size_t lds = mmvq_fast_plan(a, num_cu(), nt / 64);
    if (!lds) return hipErrorInvalidValue;
    const int blocks = a.seg_block0[3];
#define FAST(KBV, NTV, FZ) hipLaunchKernelGGL((mmvq_fast_kernel<KBV, NTV, FZ>), dim3(blocks), dim3(NTV), lds, st, a)
#define FAST_F(KBV, NTV) do { if (a.fuse_mode == 0) FAST(KBV, NTV, 0); else if (a.fuse_mode == 1) FAST(KBV, NTV, 1); else FAST(KBV, NTV, 2); } while (0)
    switch (kb) {
        case 1: if (nt == 768) FAST_F(1, 768); else FAST_F(1, 512); break;
        case 2: if (nt == 768) FAST_F(2, 768); else FAST_F(2, 512); break;
        case 3: FAST_F(3, 512); break;
        case 4: FAST_F(4, 512); break;
#define FAST_W(KBV) case KBV: if (a.fuse_mode == 2) FAST(KBV, 512, 2); else FAST(KBV, 512, 0); break
        FAST_W(5); FAST_W(6); FAST_W(7); FAST_W(9); FAST_W(10); FAST_W(11); FAST_W(14); FAST_W(15);
#undef FAST_W
        default: return hipErrorInvalidValue;
    }
#undef FAST_F
#undef FAST
    return hipGetLastError();
}

}  // namespace mi355
