/* abi_check.c -- TEST-ONLY: include/esac_hip.h is a plain C header and libesac_hip.so a plain C ABI.
 * Built with gcc -std=c99 (no C++, no HIP headers) by tests/test_abi_and_api.py; dlopens the library, resolves every
 * entry point the header declares through its C prototype, checks the parameter block's size and runs the calls that
 * need no GPU.  With a device (argv[2] == "gpu") it also runs one whole forward call on raw hipMalloc'ed buffers
 * obtained through the HIP runtime's C entry points -- no torch anywhere. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/esac_hip.h"

#define RESOLVE(name)                                              \
    do {                                                           \
        *(void**)(&p_##name) = dlsym(lib, #name);                  \
        if (!p_##name) {                                           \
            fprintf(stderr, "missing symbol %s\n", #name);         \
            return 2;                                              \
        }                                                          \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);  /* global: the HIP runtime it depends on becomes visible too */
    if (!lib) {
        fprintf(stderr, "dlopen failed: %s\n", dlerror());
        return 1;
    }
    int (*p_esac_hip_abi_version)(void);
    const char* (*p_esac_hip_last_error)(void);
    int (*p_esac_hip_device_count)(void);
    int (*p_esac_hip_create)(esac_hip_ctx**, int);
    int (*p_esac_hip_destroy)(esac_hip_ctx*);
    int (*p_esac_hip_forward)(esac_hip_ctx*, const float*, const int64_t*, const esac_hip_params*, void*, double*, double*, double*);
    int (*p_esac_hip_forward_batch)(esac_hip_ctx*, int, const float*, int64_t, const int64_t*, const esac_hip_params*, void*, double*,
                                    double*, double*);
    int (*p_esac_hip_backward)(esac_hip_ctx*, const float*, float*, const int64_t*, const float*, float, float, float,
                               const esac_hip_params*, void*, double*);
    int (*p_esac_hip_read)(esac_hip_ctx*, int, void*, size_t);
    int (*p_esac_hip_host_turn)(esac_hip_ctx*, double*);
    RESOLVE(esac_hip_abi_version);
    RESOLVE(esac_hip_last_error);
    RESOLVE(esac_hip_device_count);
    RESOLVE(esac_hip_create);
    RESOLVE(esac_hip_destroy);
    RESOLVE(esac_hip_forward);
    RESOLVE(esac_hip_forward_batch);
    RESOLVE(esac_hip_backward);
    RESOLVE(esac_hip_read);
    RESOLVE(esac_hip_host_turn);
    if (p_esac_hip_abi_version() != ESAC_HIP_ABI_VERSION) return 3;
    if (sizeof(esac_hip_params) != 104) {
        fprintf(stderr, "sizeof(esac_hip_params) = %zu\n", sizeof(esac_hip_params));
        return 4;
    }
    printf("abi %d, params %zu bytes, devices %d\n", p_esac_hip_abi_version(), sizeof(esac_hip_params), p_esac_hip_device_count());
    esac_hip_ctx* ctx = NULL;
    int rc = p_esac_hip_create(&ctx, 0);
    if (argc < 3 || strcmp(argv[2], "gpu") != 0) {
        /* no device here: creation must fail loudly, never fall back */
        if (rc == 0) return 5;
        printf("create without a device: status %d, \"%s\"\n", rc, p_esac_hip_last_error());
        return strstr(p_esac_hip_last_error(), "no CPU fallback") ? 0 : 6;
    }
    if (rc != 0) {
        fprintf(stderr, "create failed: %s\n", p_esac_hip_last_error());
        return 7;
    }
    /* device buffers through the HIP runtime's own C ABI */
    int (*hipMalloc_)(void**, size_t);
    int (*hipMemcpy_)(void*, const void*, size_t, int);
    int (*hipFree_)(void*);
    *(void**)(&hipMalloc_) = dlsym(RTLD_DEFAULT, "hipMalloc");
    *(void**)(&hipMemcpy_) = dlsym(RTLD_DEFAULT, "hipMemcpy");
    *(void**)(&hipFree_) = dlsym(RTLD_DEFAULT, "hipFree");
    if (!hipMalloc_ || !hipMemcpy_ || !hipFree_) return 9;
    /* a flat wall 3 m in front of an identity camera: scene point of cell (x,y) = ray through its pixel centre at z = 3 */
    enum { H = 60, W = 80, N = 32, P = H * W };
    float* sc = (float*)malloc(sizeof(float) * 3 * P);
    int64_t assign[N];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float px = (float)(x * 8 + 4), py = (float)(y * 8 + 4);
            sc[0 * P + y * W + x] = (px - 320.0f) / 525.0f * 3.0f;
            sc[1 * P + y * W + x] = (py - 240.0f) / 525.0f * 3.0f;
            sc[2 * P + y * W + x] = 3.0f;
        }
    for (int i = 0; i < N; i++) assign[i] = 0;
    void *d_sc = NULL, *d_assign = NULL;
    if (hipMalloc_(&d_sc, sizeof(float) * 3 * P) || hipMalloc_(&d_assign, sizeof(assign))) return 10;
    if (hipMemcpy_(d_sc, sc, sizeof(float) * 3 * P, 1) || hipMemcpy_(d_assign, assign, sizeof(assign), 1)) return 11;
    esac_hip_params p;
    memset(&p, 0, sizeof(p));
    p.E = 1; p.H = H; p.W = W; p.N = N;
    p.focal = 525.0f; p.ppx = 320.0f; p.ppy = 240.0f;
    p.inlier_thresh = 10.0f; p.inlier_alpha = 100.0f; p.inlier_beta = 0.5f; p.max_reproj = 100.0f;
    p.sub_sampling = 8; p.seed = 1305; p.call = 0; p.max_ref_steps = -1;
    double res[ESAC_RES_DOUBLES];
    rc = p_esac_hip_forward(ctx, (const float*)d_sc, (const int64_t*)d_assign, &p, NULL, NULL, NULL, res);
    if (rc != 0) {
        fprintf(stderr, "forward failed: %s\n", p_esac_hip_last_error());
        return 12;
    }
    /* the map was generated with the identity pose: the estimate must be the identity transform */
    double worst = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            const double d = res[ESAC_RES_POSE + 4 * i + j] - (i == j ? 1.0 : 0.0);
            worst = d < 0 ? (-d > worst ? -d : worst) : (d > worst ? d : worst);
        }
    printf("forward ok: expert %d, inliers %d, |pose - I|max %.2e\n", (int)res[ESAC_RES_EXPERT], (int)res[ESAC_RES_INLIERS], worst);
    /* training path: ground truth = identity -> loss ~ 0 */
    float gt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    void* d_grad = NULL;
    if (hipMalloc_(&d_grad, sizeof(float) * 3 * P)) return 13;
    memset(sc, 0, sizeof(float) * 3 * P);
    if (hipMemcpy_(d_grad, sc, sizeof(float) * 3 * P, 1)) return 14;
    double out[4];
    rc = p_esac_hip_backward(ctx, (const float*)d_sc, (float*)d_grad, (const int64_t*)d_assign, gt, 1.0f, 100.0f, 100.0f, &p, NULL, out);
    if (rc != 0) {
        fprintf(stderr, "backward failed: %s\n", p_esac_hip_last_error());
        return 15;
    }
    printf("backward ok: expected loss %.3e over %d refined hypotheses\n", out[0], (int)out[1]);
    if (argc > 3 && strcmp(argv[3], "time") == 0) {
        /* where the host's time of a blocking call goes WITHOUT Python or torch in the process (scripts/dev/host_turn.py is
         * the same split under them): mean over 300 calls */
        double acc[6] = {0, 0, 0, 0, 0, 0}, st[8];
        for (int i = 0; i < 340; i++) {
            p.call = (uint64_t)(100 + i);
            if (p_esac_hip_forward(ctx, (const float*)d_sc, (const int64_t*)d_assign, &p, NULL, NULL, NULL, res) != 0) return 17;
            p_esac_hip_host_turn(ctx, st);
            if (i >= 40)
                for (int k = 0; k < 6; k++) acc[k] += st[k];
        }
        printf("host turn, plain C caller (us): args %.2f | launch sample %.2f | score %.2f | refine %.2f | record landed after %.2f | returned %.2f\n",
               acc[0] / 300e3, (acc[1] - acc[0]) / 300e3, (acc[2] - acc[1]) / 300e3, (acc[3] - acc[2]) / 300e3, acc[4] / 300e3, acc[5] / 300e3);
    }
    if (argc > 3 && strcmp(argv[3], "comm") == 0) {
        /* the multi-GPU exchange without torch in the process: RCCL is bound by the library at this first call (dlopen of
         * librccl.so.1 = ROCm's own copy here), a one-rank communicator, one all-reduce(SUM) in place: the values stay */
        int (*p_esac_hip_comm_unique_id)(void*, size_t);
        int (*p_esac_hip_comm_init)(esac_hip_ctx*, int, int, const void*, size_t);
        int (*p_esac_hip_comm_destroy)(esac_hip_ctx*);
        int (*p_esac_hip_allreduce_sum)(esac_hip_ctx*, double*, size_t, void*);
        int (*p_esac_hip_comm_info)(esac_hip_ctx*, int32_t*);
        RESOLVE(esac_hip_comm_info);
        RESOLVE(esac_hip_comm_unique_id);
        RESOLVE(esac_hip_comm_init);
        RESOLVE(esac_hip_comm_destroy);
        RESOLVE(esac_hip_allreduce_sum);
        double h[288], back[288];
        void* d_buf = NULL;
        for (int i = 0; i < 288; i++) h[i] = 0.25 * i - 7.0;
        if (hipMalloc_(&d_buf, sizeof(h)) != 0 || hipMemcpy_(d_buf, h, sizeof(h), 1) != 0) return 18;
        if (p_esac_hip_allreduce_sum(ctx, (double*)d_buf, 288, NULL) != -13) return 19; /* no communicator yet */
        unsigned char id[ESAC_COMM_ID_BYTES];
        if (p_esac_hip_comm_unique_id(id, sizeof(id)) != 0 || p_esac_hip_comm_init(ctx, 1, 0, id, sizeof(id)) != 0) {
            fprintf(stderr, "communicator: %s\n", p_esac_hip_last_error());
            return 20;
        }
        int32_t info[4] = {-9, -9, -9, -9};
        if (p_esac_hip_comm_info(ctx, info) != 0 || info[0] != 1 || info[1] != 0 || info[3] != 0 || (info[2] != 0 && info[2] != -1)) {
            fprintf(stderr, "comm info: %d %d %d %d (%s)\n", info[0], info[1], info[2], info[3], p_esac_hip_last_error());
            return 25;
        }
        for (int rep = 0; rep < 3; rep++)
            if (p_esac_hip_allreduce_sum(ctx, (double*)d_buf, 288, NULL) != 0) {
                fprintf(stderr, "all-reduce: %s\n", p_esac_hip_last_error());
                return 21;
            }
        if (hipMemcpy_(back, d_buf, sizeof(back), 2) != 0) return 22; /* blocking copy on the NULL stream: after the collective */
        for (int i = 0; i < 288; i++)
            if (back[i] != h[i]) return 23;
        if (p_esac_hip_comm_destroy(ctx) != 0 || p_esac_hip_allreduce_sum(ctx, (double*)d_buf, 288, NULL) != -13) return 24;
        if (p_esac_hip_comm_info(ctx, info) != -13) return 26;
        hipFree_(d_buf);
        printf("comm ok: one-rank RCCL communicator, all-reduce of 288 doubles in place\n");
    }
    if (argc > 3 && strcmp(argv[3], "spec") == 0) {
        /* several experts from plain C: the wall as expert 0, two maps of noise as experts 1 and 2, 512 hypotheses of which every
         * third sits on a wrong expert.  By default the call takes the speculative route (streams of the context's own beside the
         * caller's); with ESAC_DEBUG_NO_SPECULATION every launch runs in stream order: the two records and the two score vectors
         * must be the same bits, and ESAC_BUF_SPEC_INFO must say which route ran. */
        int (*p_esac_hip_set_debug)(esac_hip_ctx*, int);
        RESOLVE(esac_hip_set_debug);
        enum { E3 = 3, N3 = 512 };
        float* sc3 = (float*)malloc(sizeof(float) * 3 * P * E3);
        int64_t* as3 = (int64_t*)malloc(sizeof(int64_t) * N3);
        unsigned lcg = 12345u;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float px = (float)(x * 8 + 4), py = (float)(y * 8 + 4);
                sc3[0 * P + y * W + x] = (px - 320.0f) / 525.0f * 3.0f;
                sc3[1 * P + y * W + x] = (py - 240.0f) / 525.0f * 3.0f;
                sc3[2 * P + y * W + x] = 3.0f;
            }
        for (int i = 3 * P; i < 3 * P * E3; i++) {
            lcg = lcg * 1664525u + 1013904223u;
            sc3[i] = (float)(lcg >> 8) / 16777216.0f * 6.0f - 3.0f + ((i / P) % 3 == 2 ? 4.0f : 0.0f); /* x, y in [-3, 3), z in [1, 7) */
        }
        for (int i = 0; i < N3; i++) as3[i] = i % 3 == 2 ? 1 + (i / 3) % 2 : 0;
        void *d_sc3 = NULL, *d_as3 = NULL, *d_scores = NULL;
        if (hipMalloc_(&d_sc3, sizeof(float) * 3 * P * E3) || hipMalloc_(&d_as3, sizeof(int64_t) * N3) || hipMalloc_(&d_scores, sizeof(double) * N3)) return 30;
        if (hipMemcpy_(d_sc3, sc3, sizeof(float) * 3 * P * E3, 1) || hipMemcpy_(d_as3, as3, sizeof(int64_t) * N3, 1)) return 31;
        esac_hip_params q = p;
        q.E = E3; q.N = N3; q.call = 77;
        double rec[2][ESAC_RES_DOUBLES];
        static double scores[2][N3];
        int32_t info[2][4];
        for (int route = 0; route < 2; route++) { /* 0: default (speculative), 1: stream order */
            if (p_esac_hip_set_debug(ctx, route ? ESAC_DEBUG_NO_SPECULATION : 0) != 0) return 32;
            if (p_esac_hip_forward(ctx, (const float*)d_sc3, (const int64_t*)d_as3, &q, NULL, (double*)d_scores, NULL, rec[route]) != 0) {
                fprintf(stderr, "forward (several experts, route %d): %s\n", route, p_esac_hip_last_error());
                return 33;
            }
            if (p_esac_hip_read(ctx, ESAC_BUF_SPEC_INFO, info[route], sizeof(info[route])) != 0) return 34;
            if (hipMemcpy_(scores[route], d_scores, sizeof(double) * N3, 2)) return 35;
        }
        p_esac_hip_set_debug(ctx, 0);
        if (!(info[0][2] == 1 && info[1][2] == 0)) {
            fprintf(stderr, "routes: speculative %d, stream order %d\n", info[0][2], info[1][2]);
            return 36;
        }
        if (memcmp(rec[0], rec[1], sizeof(double) * 31) != 0 || memcmp(scores[0], scores[1], sizeof(scores[0])) != 0) return 37;
        if ((int)rec[0][ESAC_RES_EXPERT] != 0) return 38; /* the winner sits on the wall */
        printf("spec ok: 3 experts x 512 hypotheses, speculative route == stream order bit for bit (winner %d on expert %d, %d inliers)\n",
               (int)rec[0][ESAC_RES_HYP], (int)rec[0][ESAC_RES_EXPERT], (int)rec[0][ESAC_RES_INLIERS]);
        hipFree_(d_sc3); hipFree_(d_as3); hipFree_(d_scores);
        free(sc3); free(as3);
    }
    hipFree_(d_grad); hipFree_(d_sc); hipFree_(d_assign);
    free(sc);
    p_esac_hip_destroy(ctx);
    return (worst < 1e-5 && (int)res[ESAC_RES_INLIERS] == P && out[0] >= 0 && out[0] < 1e-2) ? 0 : 16;
}
