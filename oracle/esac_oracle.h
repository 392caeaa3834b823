/*
 * esac_oracle.h -- CPU oracle for the ESAC hypothesis/inlier hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under esac_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (vislearn/esac, code/esac/esac.cpp) cannot be
 * built in this image (needs OpenCV 3.4.2: opencv_core, opencv_calib3d) and it
 * ships no tests or golden vectors.  This file restates
 *   - the reference's own control flow (esac.cpp:64-190, esac_util.h) line by
 *     line, and
 *   - the OpenCV routines it calls (solvePnP P3P / ITERATIVE, projectPoints,
 *     Rodrigues, Mat::inv) from their published algorithms, FROM MEMORY.
 * See oracle/README.md for the list of known deviations.
 */
#ifndef ESAC_ORACLE_H
#define ESAC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESAC_ORACLE_MAX_REF_STEPS 100   /* esac.cpp:45  MAX_REF_STEPS */
#define ESAC_ORACLE_MAX_TRIES 1000000   /* esac.cpp:44  MAX_SAMPLING_TRIES */

/* RNG modes. 0 = counter-based Philox4x32-10 keyed (seed, call, hyp, try, block):
 * shared with the HIP path so both draw identical minimal sets.
 * 1 = sequential callback stream (used to replay the reference's single-thread
 * mt19937 stream when validating against oracle/_ref). */
#define ESAC_RNG_PHILOX 0
#define ESAC_RNG_CALLBACK 1

typedef int (*esac_oracle_irand_fn)(int lo_incl, int hi_excl, void* user);

typedef struct esac_oracle_args {
    /* inputs (esac.cpp:64-77) */
    const float* scene_coords;   /* [E,3,H,W], element strides below            */
    int64_t sc_stride[4];
    int E, H, W;
    const int64_t* hyp_assign;   /* [N], element stride below (0 = expand())    */
    int64_t assign_stride;
    int N;
    int shift_x, shift_y;
    float focal, ppx, ppy;
    float inlier_thresh, inlier_alpha, inlier_beta, max_reproj;
    int sub_sampling;
    /* RNG key */
    uint64_t seed, call;
    int rng_mode;
    esac_oracle_irand_fn irand_cb;
    void* irand_user;
    /* limits (reference constants by default) */
    int max_tries;        /* <=0 -> ESAC_ORACLE_MAX_TRIES     */
    int max_ref_steps;    /* <0  -> ESAC_ORACLE_MAX_REF_STEPS */
    int num_threads;      /* <=0 -> omp default               */
    /* outputs (any may be NULL) */
    float*   out_pose;        /* [16] row-major 4x4, esac.cpp:182-187           */
    int32_t* out_sample_xy;   /* [N,4,2] sampled cells (x,y)                    */
    int32_t* out_tries;       /* [N] accepted try index, -1 if budget exhausted */
    double*  out_hyps;        /* [N,6] rvec,tvec before refinement              */
    double*  out_scores;      /* [N]                                            */
    double*  out_probs;       /* [N] softmax                                    */
    double*  out_entropy;     /* [1]                                            */
    int32_t* out_winner;      /* [1] hypothesis index                           */
    double*  out_refined;     /* [6] rvec,tvec after refinement                 */
    int32_t* out_ref_steps;   /* [1] number of accepted refinement re-fits      */
    int32_t* out_inlier_counts; /* [max_ref_steps+1] inlier count seen at each step */
    uint8_t* out_inlier_map;  /* [H*W] row-major (y,x), last accepted inlier set */
    float*   out_winner_errs; /* [H*W] reprojection error image of the winner (pre-refinement) */
    double*  out_phase_ms;    /* [4] sampling, scoring, selection, refinement   */
    int32_t* out_lm_iters;    /* [1] total LM iterations spent in refinement    */
    /* optional [N]: global index of each hypothesis (keys its RNG stream); NULL -> 0..N-1.
     * Lets a shard of a larger problem be evaluated on its own (multi-GPU tests). */
    const int32_t* hyp_index;
    /* optional [N,6]: hypotheses (rvec,tvec) to use INSTEAD of sampling (tests that replay the reference's own
     * hypotheses, or place a pose by hand); out_tries is then all 0 and out_sample_xy all 0. */
    const double* in_hyps;
} esac_oracle_args;

/* returns winning expert (>=0) or <0 on argument error */
int esac_oracle_forward(esac_oracle_args* a);

/* esac_backward (esac.cpp:213-511): expected pose loss + its gradient wrt the scene coordinates.
 * `fwd` carries the inputs shared with the forward pass (its out_* members are ignored). */
typedef struct esac_oracle_bwd_args {
    esac_oracle_args fwd;
    const float* gt_pose;      /* [16] row-major 4x4 camera pose (gtPose, esac.cpp:217)            */
    float w_rot, w_trans, loss_cut;
    float* out_gradients;      /* [E,3,H,W] accumulated with += (esac.cpp:501-506), strides below  */
    int64_t grad_stride[4];
    /* optional stage outputs */
    double* out_probs;         /* [N] */
    double* out_losses;        /* [N] */
    double* out_init_hyps;     /* [N,6] */
    double* out_ref_hyps;      /* [N,6] */
    double* out_score_grads;   /* [N] scoreOutputGradients (esac_derivative.h:368-375) */
    double* out_dloss;         /* [N,6] dLoss per refined hypothesis */
    int32_t* out_sample_xy;    /* [N,4,2] */
    double* out_entropy;       /* [1] */
    double* out_grad_path1;    /* [N,H*W,3] gradients[h] before the probability weight (esac.cpp:456-463) */
    double* out_grad_path2;    /* [N,H*W,3] dLoss_dScore_dObjs[h] (esac.cpp:472-486) */
} esac_oracle_bwd_args;

/* returns the expected loss (>= 0) or -1 on argument error */
double esac_oracle_backward(esac_oracle_bwd_args* b);

/* ---- building blocks exported for known-answer tests ---- */
void esac_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void esac_oracle_draw_cells(uint64_t seed, uint64_t call, uint32_t hyp, uint32_t tr,
                            int W, int H, int32_t xy[8]);
/* obj: 4x3 doubles, img: 4x2 doubles (pixels). returns 1 on success */
int  esac_oracle_p3p(const double* obj, const double* img, double fx, double fy,
                     double cx, double cy, double rvec[3], double tvec[3]);
/* all solutions from the first 3 points; R [4][9], t [4][3]; returns count */
int  esac_oracle_p3p_all(const double* obj3, const double* img3, double fx, double fy,
                         double cx, double cy, double* R, double* t);
int  esac_oracle_solve_deg4(double a, double b, double c, double d, double e, double roots[4]);
void esac_oracle_rodrigues_vec2mat(const double r[3], double R[9], double dRdr[27]);
void esac_oracle_rodrigues_mat2vec(const double R[9], double r[3]);
void esac_oracle_rodrigues_mat2vec_svd(const double R[9], double r[3]); /* with OpenCV's U*Vt re-orthonormalisation */
void esac_oracle_project(const double rvec[3], const double tvec[3], double fx, double fy,
                         double cx, double cy, const float* pts3, int n, float* uv);
/* LM refit (cv::solvePnP ITERATIVE, useExtrinsicGuess=true). pose in/out (rvec,tvec).
 * returns number of outer LM iterations */
int  esac_oracle_lm_pnp(const float* obj, const float* img, int n, double fx, double fy,
                        double cx, double cy, double pose[6]);
void esac_oracle_pose2trans(const double pose[6], double T[16]);
void esac_oracle_inv4(const double M[16], double Minv[16]);
void esac_oracle_pinv_sym6(const double A[36], double Ainv[36]);
/* single routines of the training path (esac_loss.h, esac_derivative.h, esac_util.h:333-351,555-568) */
double esac_oracle_pose_loss(const double pose[6], const double gt_trans[16], double wRot, double wTrans, double cut);
void esac_oracle_pose_dloss(const double est[6], const double gt_pose[6], double wRot, double wTrans, double cut, double jac[6]);
void esac_oracle_trans2pose(const double T[16], double pose[6]);
void esac_oracle_dproject_dobj(float ptx, float pty, float ox, float oy, float oz, const double rvec[3], const double t[3],
                               float focal, float ppx, float ppy, float maxReproj, double out[3]);
int esac_oracle_norm_jac_row(const double rvec[3], const double t[3], float focal, float ppx, float ppy, float X, float Y,
                             float Z, float px, float py, float maxReproj, double row[6]);
void esac_oracle_project_jac(const double rvec[3], const double tvec[3], double fx, double fy, double cx, double cy,
                             const float* pts3, int n, double* J12);
int  esac_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
