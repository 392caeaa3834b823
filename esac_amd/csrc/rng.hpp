// rng.hpp -- counter-based sampling stream for hypothesis minimal sets.
//
// The reference draws cells with one std::mt19937 per OpenMP thread whose state
// persists across calls (thread_rand.cpp:13-42): the draws of hypothesis h
// depend on the thread count and on every earlier call.  A GPU cannot and
// should not reproduce that; instead try t of hypothesis h of call c reads a
// Philox4x32-10 stream keyed (seed, c) at counter (h, t, block): any lane can
// evaluate any try, and a wavefront testing 64 tries in lock-step accepts
// exactly the try a sequential loop would.
// Range semantics are the reference's: irand(0, imW-1) has an exclusive upper
// bound (thread_rand.cpp:68-71), so x is uniform on [0, W-2], y on [0, H-2].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esac {

struct Philox {
    uint32_t k0, k1;
    __host__ __device__ Philox(uint64_t seed, uint64_t call) {
        const uint64_t key = seed + call * 0x9E3779B97F4A7C15ull;
        k0 = (uint32_t)key;
        k1 = (uint32_t)(key >> 32);
    }
    __device__ __forceinline__ void block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) const {
        uint32_t ka = k0, kb = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            c0 = hi1 ^ c1 ^ ka;
            c1 = lo1;
            c2 = hi0 ^ c3 ^ kb;
            c3 = lo0;
            ka += 0x9E3779B9u;
            kb += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// The four DISTINCT cells of try (hyp, tr): candidate cells 2k, 2k+1 come from
// stream block k; a candidate equal to an already chosen cell is skipped, which
// is the reference's `j--; continue` redraw (esac_util.h:170-174).
__device__ __forceinline__ void draw_cells(const Philox& rng, uint32_t hyp, uint32_t tr, int W, int H, int cx[4],
                                           int cy[4]) {
    int have = 0;
    const uint32_t nx = (uint32_t)(W - 1), ny = (uint32_t)(H - 1);
    for (uint32_t k = 0; have < 4; k++) {
        uint32_t o[4];
        rng.block(hyp, tr, k, 0x45534143u, o);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int x = (int)__umulhi(o[2 * half], nx);
            const int y = (int)__umulhi(o[2 * half + 1], ny);
            bool dup = false;
#pragma unroll
            for (int j = 0; j < 4; j++) dup |= (j < have) && cx[j] == x && cy[j] == y;
            if (!dup && have < 4) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j == have) {
                        cx[j] = x;
                        cy[j] = y;
                    }
                have++;
            }
        }
    }
}

}  // namespace esac
