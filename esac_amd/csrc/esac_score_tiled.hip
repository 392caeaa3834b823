// esac_score_tiled.hip -- K2 for large maps / many hypotheses per expert: the soft-inlier score with the map TILE
// stationary and the hypotheses streaming past it.
//
// Reference: getReproErrs + getHypScores per hypothesis over all cells, esac.cpp:131-147, esac_util.h:235-363.
// The one-hypothesis-per-workgroup kernel (k_score_fast) re-reads the whole map per hypothesis: 12*H*W bytes each,
// served by L2 while a map fits there (57.6 KB at 60x80).  A 480x640 map is 3.7 MB per expert -- 50 experts x 16384
// hypotheses pull 60 GB through the cache hierarchy (11 GB reached the fabric in the round-1 profile) for 184 MB of
// distinct data.  Here the loop nest is turned inside out:
//   1. k_bucket_{count,scan,scatter}: counting sort of the hypotheses by expert (the histogram test_esac.py:178 also
//      takes), a table of chunks (expert, first position, count <= TILE_HC) and the fp32 poses copied into sorted order;
//   2. k_score_tiled: ONE WAVEFRONT owns (chunk, sub-tile of 768 cells): it loads its 768 cells once -- 16-byte
//      coalesced loads, 12 cells per lane kept in registers together with their pixel positions -- and walks the
//      chunk's hypotheses: pose through scalar loads into SGPRs (the next pose is fetched while the current one is
//      evaluated), ~22 fp32 VALU ops per cell, DPP/permlane wavefront reduction, one partial sum per (hypothesis,
//      sub-tile); 64 partials are gathered across the lanes and stored with one coalesced 256-byte write;
//   3. k_score_tiled_reduce: sum of a hypothesis' partials over the sub-tiles in fixed order (deterministic).
// Every map byte is read from HBM once per chunk row instead of once per hypothesis; sub-tile s is handled on XCD
// s % 8 by all of its chunks, so the re-reads of a tile by the chunks of a large expert hit that XCD's L2.
// No LDS staging: a sub-tile lives in the registers of the wavefront that owns it for the whole hypothesis loop,
// there is nothing to share between wavefronts.  No MFMA: per cell a 3x4 transform, a division and three
// transcendentals.  Bound: fp32 VALU issue (256 CUs x 4 SIMD-32 at 2.4 GHz), see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"
#include "esac_kernels.hpp"

namespace esac {

constexpr int TILE_CPT = 12;     // cells per lane (a multiple of 4)
constexpr int TILE_CELLS = 64 * TILE_CPT;   // cells per sub-tile (one wavefront)
constexpr int BUCKET_B = 1024;

// ---------------------------------------------------------------- 1. bucket the hypotheses by expert
// Counting sort in three small launches (one 1024-thread workgroup walking all N hypotheses with dependent loads and LDS
// atomics took 66 us at N = 16384):
//   k_bucket_count    every workgroup histograms its 1024 hypotheses in LDS and adds the non-zero bins to counts[E];
//   k_bucket_scan     one workgroup: exclusive scans of the counts and of the chunk counts -> first sorted position of
//                     every expert (start[e], also the running fill pointer of the scatter) and the chunk table;
//   k_bucket_scatter  pos = atomicAdd(fill[e], 1): order[pos] = h and the pose in the form the tile kernel consumes.
// The order inside an expert is whatever the atomics produce -- nothing downstream depends on it: every hypothesis'
// partial sums are its own (k_score_tiled_reduce scatters them back through `order`).
__global__ __launch_bounds__(BUCKET_B) void k_bucket_count(KArgs a) {
    __shared__ int s_cnt[ESAC_TILED_MAX_EXPERTS];
    const int E = a.E;
    for (int e = threadIdx.x; e < E; e += BUCKET_B) s_cnt[e] = 0;
    __syncthreads();
    const int h = blockIdx.x * BUCKET_B + threadIdx.x;
    if (h < a.N) atomicAdd(&s_cnt[expert_of(a, h)], 1);
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += BUCKET_B)
        if (s_cnt[e]) atomicAdd(a.bucket_fill + e, s_cnt[e]);
}

__global__ __launch_bounds__(BUCKET_B) void k_bucket_scan(KArgs a) {
    __shared__ int s_scan[2][BUCKET_B];
    const int E = a.E, HC = ESAC_TILED_HC;
    // every thread owns EPT consecutive experts, the per-thread totals are scanned across the workgroup (Hillis-Steele
    // in LDS), once for the counts and once for the chunk counts
    constexpr int EPT = ESAC_TILED_MAX_EXPERTS / BUCKET_B;
    int cnt[EPT];
    int mine_cnt = 0, mine_chunks = 0;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = threadIdx.x * EPT + k;
        cnt[k] = e < E ? a.bucket_fill[e] : 0;
        mine_cnt += cnt[k];
        mine_chunks += (cnt[k] + HC - 1) / HC;
    }
    int excl[2];
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const int mine = which == 0 ? mine_cnt : mine_chunks;
        int cur = 0;
        s_scan[0][threadIdx.x] = mine;
        __syncthreads();
        for (int off = 1; off < BUCKET_B; off <<= 1) {
            const int v = s_scan[cur][threadIdx.x] + ((int)threadIdx.x >= off ? s_scan[cur][threadIdx.x - off] : 0);
            s_scan[cur ^ 1][threadIdx.x] = v;
            cur ^= 1;
            __syncthreads();
        }
        excl[which] = s_scan[cur][threadIdx.x] - mine;
        if (which == 1 && threadIdx.x == BUCKET_B - 1) a.n_chunks[0] = s_scan[cur][threadIdx.x];
        __syncthreads();
    }
    int pos = excl[0], ch = excl[1];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = threadIdx.x * EPT + k;
        if (e < E) {
            const int c = cnt[k];
            a.bucket_fill[e] = pos;  // from here on: where the next hypothesis of expert e goes
            for (int j = 0, first = 0; first < c; j++, first += HC) {
                int* row = a.chunks + 4 * (size_t)(ch + j);
                row[0] = e;
                row[1] = pos + first;
                row[2] = c - first < HC ? c - first : HC;
                row[3] = 0;
            }
            pos += c;
            ch += (c + HC - 1) / HC;
        }
    }
}

__global__ __launch_bounds__(BUCKET_B) void k_bucket_scatter(KArgs a) {
    // ranks inside the workgroup through LDS, ONE global atomic per (workgroup, expert) to reserve the range: 16384 global
    // atomics on 50 addresses serialise (45 us measured); 16 workgroups x 50 experts do not
    __shared__ int s_cnt[ESAC_TILED_MAX_EXPERTS];
    __shared__ int s_base[ESAC_TILED_MAX_EXPERTS];
    const int E = a.E;
    for (int e = threadIdx.x; e < E; e += BUCKET_B) s_cnt[e] = 0;
    __syncthreads();
    const int h = blockIdx.x * BUCKET_B + threadIdx.x;
    int e = 0, local = 0;
    if (h < a.N) {
        e = expert_of(a, h);
        local = atomicAdd(&s_cnt[e], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < E; k += BUCKET_B)
        if (s_cnt[k]) s_base[k] = atomicAdd(a.bucket_fill + k, s_cnt[k]);
    __syncthreads();
    if (h >= a.N) return;
    const int pos = s_base[e] + local;
    a.order[pos] = h;
    // the pose in the form the tile kernel consumes: camera folded into the rows, rows A and B and (in the kernel) the
    // pixel positions scaled by |beta| log2(e) so that the exponent of the sigmoid needs no arithmetic of its own (see PoseU)
    const float4* src = reinterpret_cast<const float4*>(a.rt32 + (size_t)h * 12);
    const float4 q0 = src[0], q1 = src[1], q2 = src[2];  // r0 r1 r2 r3 | r4 r5 r6 r7 | r8 t0 t1 t2
    const float kb = fabsf(a.beta) * 1.4426950408889634f;
    const float f = a.focal * kb, cx = a.ppx * kb, cy = a.ppy * kb;
    float4* dst = reinterpret_cast<float4*>(a.rt_sorted + (size_t)pos * 12);
    dst[0] = make_float4(fmaf(f, q0.x, cx * q1.z), fmaf(f, q0.y, cx * q1.w), fmaf(f, q0.z, cx * q2.x), fmaf(f, q2.y, cx * q2.w));
    dst[1] = make_float4(fmaf(f, q0.w, cy * q1.z), fmaf(f, q1.x, cy * q1.w), fmaf(f, q1.y, cy * q2.x), fmaf(f, q2.z, cy * q2.w));
    dst[2] = make_float4(q1.z, q1.w, q2.x, q2.w);
}

// ---------------------------------------------------------------- 2. the tile-stationary score
// Pose rows with the camera folded in: u = (A . X + ta) / (C . X + tc),  v = (B . X + tb) / (C . X + tc),
// A = f R0 + cx R2, B = f R1 + cy R2, C = R2 (likewise the translation): two FMAs per cell less than projecting first
// and applying (f, c) afterwards.  Rows A and B and the pixel positions also carry the factor k = |beta| log2(e): the
// distance the kernel forms is k * err, the exponent of the sigmoid 2^(k err - k tau) needs no multiply-add of its own
// and the constant 2^(-k tau) rides in the FMA that forms 1 + exp.  k_bucket_scatter stores the rows in this form; the
// tile kernel keeps them in SGPRs.
struct PoseU {
    float a0, a1, a2, ta, b0, b1, b2, tb, c0, c1, c2, tc;
};
typedef float sgpr4 __attribute__((ext_vector_type(4)));
// 48 bytes at a wave-uniform address through the scalar cache, asynchronously: the values may only be touched after
// pose_wait().  (Plain C++ loads end up as vector loads here -- the pointers come out of a by-value struct, so the
// compiler cannot prove the memory read-only -- and the pose would occupy 12 VGPRs instead of SGPR operands.)
__device__ __forceinline__ void pose_issue(const float* __restrict__ p, sgpr4& q0, sgpr4& q1, sgpr4& q2) {
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20"
                 : "=&s"(q0), "=&s"(q1), "=&s"(q2)
                 : "s"(p)
                 : "memory");
}
__device__ __forceinline__ void pose_wait(sgpr4& q0, sgpr4& q1, sgpr4& q2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1), "+s"(q2));
}
__device__ __forceinline__ PoseU pose_of(const sgpr4& q0, const sgpr4& q1, const sgpr4& q2) {
    return PoseU{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
}

// one cell: 1 / (1 + exp(beta (min(err, maxReproj) - tau))), err = |pixel - projection|.  Everything arrives scaled by
// k = |beta| log2(e) (rows A, B, px, py, kmax = k maxReproj); c0 = 2^(-k tau); NEG: beta < 0 (the exponent changes sign).
// two cells at a time with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: 4.9 cycles for two operations against
// 3.0 for one, scripts/dev/valu_rate.hip); the four transcendentals and the clamp have no packed form
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
template <bool NEG>
__device__ __forceinline__ f32x2 soft_inlier_tile2(const PoseU& p, f32x2 X, f32x2 Y, f32x2 Z, f32x2 px, float py, float kmax, float c0) {
    const f32x2 un = __builtin_elementwise_fma(splat2(p.a0), X, __builtin_elementwise_fma(splat2(p.a1), Y, __builtin_elementwise_fma(splat2(p.a2), Z, splat2(p.ta))));
    const f32x2 vn = __builtin_elementwise_fma(splat2(p.b0), X, __builtin_elementwise_fma(splat2(p.b1), Y, __builtin_elementwise_fma(splat2(p.b2), Z, splat2(p.tb))));
    const f32x2 zc = __builtin_elementwise_fma(splat2(p.c0), X, __builtin_elementwise_fma(splat2(p.c1), Y, __builtin_elementwise_fma(splat2(p.c2), Z, splat2(p.tc))));
    // err = |pixel - projection| = sqrt(d2n) / |zc| with d2n = (px zc - un)^2 + (py zc - vn)^2, formed as d2n * rsq(d2n zc^2):
    // ONE transcendental (v_rsq_f32) where 1/zc and the square root were two -- 3 per cell with the sigmoid's exp2 and rcp
    // (k_score_tiled 2.71 -> 2.65 ms, k_score_fast<256> 74.8 -> 68.5 us, same box).  The sign of zc drops out (no
    // cheirality test, as in the reference); + 1e-36 keeps d2n = 0 at err = 0 and sends zc = 0 to the maxReproj clamp; a NaN
    // anywhere ends in fminf(NaN, kmax) = kmax.
    const f32x2 da = __builtin_elementwise_fma(px, zc, -un);
    const f32x2 db = __builtin_elementwise_fma(splat2(py), zc, -vn);
    const f32x2 d2n = __builtin_elementwise_fma(da, da, db * db);
    const f32x2 q = __builtin_elementwise_fma(d2n, zc * zc, splat2(1e-36f));
    // (q - q: 0, or NaN when d2n zc^2 overflowed -- a hypothesis far beyond the scene: rsq(inf) = 0 would read err = 0, a perfect
    // inlier; the NaN ends in the clamp below like the rcp + sqrt form did.  One packed add per pair of cells.)
    const f32x2 er = __builtin_elementwise_fma(d2n, f32x2{__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)}, q - q);
    const f32x2 err = {fminf(er.x, kmax), fminf(er.y, kmax)};
    const f32x2 ex = {__builtin_amdgcn_exp2f(NEG ? -err.x : err.x), __builtin_amdgcn_exp2f(NEG ? -err.y : err.y)};
    const f32x2 den = __builtin_elementwise_fma(ex, splat2(c0), splat2(1.0f));
    return f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
}

template <bool NEG>
__global__ __launch_bounds__(64) void k_score_tiled(KArgs a) {
    // block -> (sub-tile, chunk): all chunks of one sub-tile are neighbours in dispatch order AND on one XCD
    // (workgroup b runs on XCD b % 8): the tile's bytes come from HBM once, the other chunks hit that XCD's L2
    const int L = blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    const int chunk = k % a.n_chunks_max;
    const int st = (k / a.n_chunks_max) * 8 + xcd;
    if (chunk >= a.n_chunks[0] || st >= a.n_sub) return;
    const int* row = a.chunks + 4 * (size_t)chunk;
    const int e = __builtin_amdgcn_readfirstlane(row[0]), first = __builtin_amdgcn_readfirstlane(row[1]),
              count = __builtin_amdgcn_readfirstlane(row[2]);
    const int lane = threadIdx.x;
    const int P = a.H * a.W;
    const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
    const Centre o = map_centre(a, mx);
    // this lane's TILE_CPT cells: groups of 4 consecutive cells (W % 4 == 0: a group never straddles a row), group g of lane l = group g * 64 + l of the sub-tile
    float X[TILE_CPT], Y[TILE_CPT], Z[TILE_CPT], px[TILE_CPT], py[TILE_CPT / 4], m[TILE_CPT / 4];  // py, weight: per group of 4
    // everything the distance is formed from carries k = |beta| log2(e) (see PoseU); e0 = 2^(-k tau) for beta > 0, 2^(k tau) else
    const float kb = fabsf(a.beta) * 1.4426950408889634f, kmax = kb * a.max_reproj;
    const float e0 = __builtin_amdgcn_exp2f(NEG ? kb * a.tau : -kb * a.tau);
    const float step = kb * (float)a.sub;
#pragma unroll
    for (int g = 0; g < TILE_CPT / 4; g++) {
        const int cell = st * TILE_CELLS + (g * 64 + lane) * 4;
        const bool valid = cell < P;            // P % 4 == 0: a group is valid or invalid as a whole
        const int cc = valid ? cell : P - 4;    // clamped, unconditional loads; weight 0 keeps them out of the sum
        const float4 vx = *reinterpret_cast<const float4*>(mx + cc);
        const float4 vy = *reinterpret_cast<const float4*>(mx + P + cc);
        const float4 vz = *reinterpret_cast<const float4*>(mx + 2 * P + cc);
        const int r = cc / a.W, c0 = cc - r * a.W;
        const float pxf = kb * cell_px(a, c0), pyf = kb * cell_py(a, r);
        X[4 * g] = vx.x - o.x; X[4 * g + 1] = vx.y - o.x; X[4 * g + 2] = vx.z - o.x; X[4 * g + 3] = vx.w - o.x;
        Y[4 * g] = vy.x - o.y; Y[4 * g + 1] = vy.y - o.y; Y[4 * g + 2] = vy.z - o.y; Y[4 * g + 3] = vy.w - o.y;
        Z[4 * g] = vz.x - o.z; Z[4 * g + 1] = vz.y - o.z; Z[4 * g + 2] = vz.z - o.z; Z[4 * g + 3] = vz.w - o.z;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            px[4 * g + l] = pxf + (float)l * step;
        }
        py[g] = pyf;
        m[g] = valid ? 1.0f : 0.0f;
    }
    const float* __restrict__ poses = a.rt_sorted + (size_t)first * 12;
    float* __restrict__ out = a.partials + (size_t)st * a.N + first;
    float res = 0.0f;
    sgpr4 c0, c1, c2, n0, n1, n2;
    pose_issue(poses, c0, c1, c2);
    pose_wait(c0, c1, c2);
    for (int i = 0; i < count; i++) {
        // the next pose's scalar loads are in flight during this hypothesis' ~200 VALU instructions (clamped: no branch)
        pose_issue(poses + (size_t)(i + 1 < count ? i + 1 : i) * 12, n0, n1, n2);
        const PoseU cur = pose_of(c0, c1, c2);
        // two cells per packed instruction: 2.80 ms against 3.67 ms for the scalar form on config 5b (round 2)
        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < TILE_CPT; u += 2)
            acc = __builtin_elementwise_fma(splat2(m[u >> 2]),
                                            soft_inlier_tile2<NEG>(cur, f32x2{X[u], X[u + 1]}, f32x2{Y[u], Y[u + 1]}, f32x2{Z[u], Z[u + 1]},
                                                                   f32x2{px[u], px[u + 1]}, py[u >> 2], kmax, e0), acc);
        const float acc0 = acc.x, acc1 = acc.y;
        const float tot = wave_sum(acc0 + acc1);  // the same total in every lane
        if ((i & 63) == lane) res = tot;
        if ((i & 63) == 63) {  // 64 partial sums gathered across the lanes: one coalesced 256-byte store
            out[i - 63 + lane] = res;
        }
        pose_wait(n0, n1, n2);
        c0 = n0; c1 = n1; c2 = n2;
    }
    const int tail = count & 63;
    if (lane < tail) out[count - tail + lane] = res;
}

// ---------------------------------------------------------------- 3. partial sums -> scores
// fast_scores[order[pos]] = alpha / W / H * sum over sub-tiles of partials[st][pos].  32 hypotheses per workgroup, eight
// lanes each: lane s adds sub-tiles s, s + 8, ... (four loads in flight), the eight partial sums are added in lane order.
// (One lane per hypothesis walked its ~400 sub-tiles in 100 dependent steps on 64 CUs: 32 us at config 5b.)
__global__ __launch_bounds__(256) void k_score_tiled_reduce(KArgs a) {
    __shared__ double s_part[8][32];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int pos = blockIdx.x * 32 + col;
    double tot = 0;
    if (pos < a.N) {
        int st = slice;
        for (; st + 24 < a.n_sub; st += 32) {  // four loads in flight
            const float p0 = a.partials[(size_t)st * a.N + pos], p1 = a.partials[(size_t)(st + 8) * a.N + pos];
            const float p2 = a.partials[(size_t)(st + 16) * a.N + pos], p3 = a.partials[(size_t)(st + 24) * a.N + pos];
            tot += (double)p0;
            tot += (double)p1;
            tot += (double)p2;
            tot += (double)p3;
        }
        for (; st < a.n_sub; st += 8) tot += (double)a.partials[(size_t)st * a.N + pos];
    }
    s_part[slice][col] = tot;
    __syncthreads();
    if (slice == 0 && pos < a.N) {
        double sum = s_part[0][col];
#pragma unroll
        for (int k = 1; k < 8; k++) sum += s_part[k][col];
        const float scale = a.alpha / a.W / a.H;  // float / int / int (esac_util.h:256)
        a.fast_scores[a.order[pos]] = (float)(sum * (double)scale);
    }
}

int tiled_sub_tiles(int P) { return (P + TILE_CELLS - 1) / TILE_CELLS; }

// order[] (hypotheses sorted by expert), chunks, rt_sorted from the assignment vector (and the current rt32 rows)
void launch_bucket_order(const KArgs& a, hipStream_t s) {
    (void)hipMemsetAsync(a.bucket_fill, 0, (size_t)a.E * sizeof(int), s);
    hipLaunchKernelGGL(k_bucket_count, dim3((a.N + BUCKET_B - 1) / BUCKET_B), dim3(BUCKET_B), 0, s, a);
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(BUCKET_B), 0, s, a);
    hipLaunchKernelGGL(k_bucket_scatter, dim3((a.N + BUCKET_B - 1) / BUCKET_B), dim3(BUCKET_B), 0, s, a);
}

void launch_score_tiled(const KArgs& a, hipStream_t s) {
    launch_bucket_order(a, s);
    const long long per_xcd = (long long)((a.n_sub + 7) / 8) * a.n_chunks_max;
    if (a.beta < 0) hipLaunchKernelGGL(k_score_tiled<true>, dim3((unsigned)(per_xcd * 8)), dim3(64), 0, s, a);
    else            hipLaunchKernelGGL(k_score_tiled<false>, dim3((unsigned)(per_xcd * 8)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_score_tiled_reduce, dim3((a.N + 31) / 32), dim3(256), 0, s, a);
}

}  // namespace esac
