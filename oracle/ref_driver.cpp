// ref_driver.cpp -- builds oracle/_ref/libesac_ref.so.  TEST INFRASTRUCTURE ONLY.
//
// Compiles the REFERENCE'S OWN sources from where they lie (-I/root/reference/code/esac):
//     esac_types.h, esac_util.h, thread_rand.h / thread_rand.cpp      (unmodified, never copied)
// against oracle/ref_shim (stand-ins for <opencv2/opencv.hpp> and at::TensorAccessor) and drives them in
// the order of esac_forward (code/esac/esac.cpp:80-189).  esac.cpp itself is not compiled: it also pulls in
// torch, esac_loss.h and esac_derivative.h (training path) -- the glue below restates only its ~40 lines of
// orchestration, every algorithmic function it calls is the reference's.
//
// What this pins: the oracle's restatement of the reference's control flow and float/double mixes
// (sampleHypotheses, getReproErrs, getHypScores, softMax, entropy, draw, refineHyp, pose2trans).
// What it does NOT pin: OpenCV's internals -- the shim forwards them to the oracle's own restatements.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

// The reference's extension source itself, UNMODIFIED, from /root/reference/code/esac (-I on the command line):
// it includes <torch/torch.h> and <opencv2/opencv.hpp> (both resolve to the stand-ins in oracle/ref_shim) and its own
// thread_rand.h, stop_watch.h, esac_types.h, esac_util.h, esac_loss.h, esac_derivative.h.
#define TORCH_EXTENSION_NAME esac
#include "esac.cpp"  // reference: esac_forward (esac.cpp:64-190), esac_backward (esac.cpp:213-511)

// single-thread replay of the reference RNG for the oracle's callback mode: ThreadRand::irand(min, max) is
// std::uniform_int_distribution<int>(min, max) on a std::mt19937 seeded 1305 + thread id (thread_rand.cpp:13-42)
static std::mt19937 g_replay;

extern "C" {

void ref_rng_reset(unsigned seed) {
    ThreadRand::forceInit(seed);  // thread_rand.cpp:7-11 (public, just not exported to Python by the reference)
    g_replay = std::mt19937();
    g_replay.seed(seed);          // generator of thread 0
}

// esac_oracle_irand_fn: (lo inclusive, hi exclusive) like the reference's irand() wrapper (thread_rand.cpp:68-71)
int ref_replay_irand(int lo, int hi_excl, void*) {
    std::uniform_int_distribution<int> dist(lo, hi_excl - 1);
    return dist(g_replay);
}

// esac_forward (esac.cpp:64-190) with the stage outputs exported. Run with OMP_NUM_THREADS=1 for a defined RNG order.
int ref_forward(const float* sc, int E, int H, int W, const int64_t* assign, int N, float* out_pose16, int shiftX,
                int shiftY, float focalLength, float ppointX, float ppointY, float inlierThreshold, float inlierAlpha,
                float inlierBeta, float maxReproj, int subSampling, unsigned maxTries, unsigned maxRefSteps,
                int32_t* out_sample_xy, double* out_hyps, double* out_scores, int32_t* out_winner, double* out_refined,
                unsigned char* out_inlier_map, double* out_entropy) {
    const int64_t sizes4[4] = {E, 3, H, W}, strides4[4] = {(int64_t)3 * H * W, (int64_t)H * W, W, 1};
    const int64_t sizes1[1] = {N}, strides1[1] = {1};
    esac::coord_t sceneCoordinates(const_cast<float*>(sc), sizes4, strides4);
    esac::hyp_assign_t hypAssignment(reinterpret_cast<long*>(const_cast<int64_t*>(assign)), sizes1, strides1);

    int imH = sceneCoordinates.size(2);
    int imW = sceneCoordinates.size(3);
    int hypCount = hypAssignment.size(0);

    cv::Mat_<float> camMat = cv::Mat_<float>::eye(3, 3);  // esac.cpp:93-97
    camMat(0, 0) = focalLength;
    camMat(1, 1) = focalLength;
    camMat(0, 2) = ppointX;
    camMat(1, 2) = ppointY;

    cv::Mat_<cv::Point2i> sampling = esac::createSampling(imW, imH, subSampling, shiftX, shiftY);  // esac.cpp:100

    std::vector<esac::pose_t> hypotheses;
    std::vector<std::vector<cv::Point2i>> sampledPoints;
    std::vector<std::vector<cv::Point2f>> imgPts;
    std::vector<std::vector<cv::Point3f>> objPts;
    esac::sampleHypotheses(sceneCoordinates, hypAssignment, sampling, camMat, maxTries, inlierThreshold, hypotheses,
                           sampledPoints, imgPts, objPts);  // esac.cpp:112

    std::vector<cv::Mat_<float>> reproErrs(hypCount);  // esac.cpp:128-140
    cv::Mat_<double> jacobeanDummy;
#pragma omp parallel for
    for (unsigned h = 0; h < hypotheses.size(); h++)
        reproErrs[h] = esac::getReproErrs(sceneCoordinates, hypotheses[h], hypAssignment[h], sampling, camMat, maxReproj,
                                          jacobeanDummy);

    std::vector<double> scores = esac::getHypScores(reproErrs, inlierThreshold, inlierAlpha, inlierBeta);  // esac.cpp:143
    std::vector<double> hypProbs = esac::softMax(scores);                                                   // esac.cpp:153
    double hypEntropy = esac::entropy(hypProbs);
    int hypIdx = esac::draw(hypProbs, false);

    for (int h = 0; h < hypCount; h++) {
        for (int j = 0; j < 4; j++) {
            out_sample_xy[8 * h + 2 * j] = sampledPoints[h][j].x;
            out_sample_xy[8 * h + 2 * j + 1] = sampledPoints[h][j].y;
        }
        for (int k = 0; k < 3; k++) {
            out_hyps[6 * h + k] = hypotheses[h].first.total() == 3 ? (hypotheses[h].first.rows == 3 ? hypotheses[h].first.at<double>(k, 0) : hypotheses[h].first.at<double>(0, k)) : 0.0;
            out_hyps[6 * h + 3 + k] = hypotheses[h].second.total() == 3 ? (hypotheses[h].second.rows == 3 ? hypotheses[h].second.at<double>(k, 0) : hypotheses[h].second.at<double>(0, k)) : 0.0;
        }
        out_scores[h] = scores[h];
    }
    *out_winner = hypIdx;
    *out_entropy = hypEntropy;

    cv::Mat_<int> inlierMap;  // esac.cpp:164-177
    esac::refineHyp(sceneCoordinates, reproErrs[hypIdx], sampling, camMat, hypAssignment[hypIdx], inlierThreshold,
                    maxRefSteps, maxReproj, hypotheses[hypIdx], inlierMap);
    for (int k = 0; k < 3; k++) {
        out_refined[k] = hypotheses[hypIdx].first.at<double>(k, 0);
        out_refined[3 + k] = hypotheses[hypIdx].second.at<double>(k, 0);
    }
    for (int y = 0; y < imH; y++)
        for (int x = 0; x < imW; x++) out_inlier_map[y * imW + x] = inlierMap.empty() ? 0 : (unsigned char)inlierMap(y, x);

    esac::trans_t estTrans = esac::pose2trans(hypotheses[hypIdx]);  // esac.cpp:182-187
    for (unsigned x = 0; x < 4; x++)
        for (unsigned y = 0; y < 4; y++) out_pose16[y * 4 + x] = estTrans(y, x);

    return hypAssignment[hypIdx];  // esac.cpp:189
}

// ---- the reference's real entry points, called exactly as pybind11 would ---------------------------------
int ref_esac_forward(float* sc, int E, int H, int W, int64_t* assign, int N, float* out_pose16, int shiftX, int shiftY,
                     float f, float ppx, float ppy, float thr, float alpha, float beta, float maxReproj, int sub) {
    at::Tensor tsc(sc, {E, 3, H, W}), tas(assign, {N}), tpose(out_pose16, {4, 4});
    return esac_forward(tsc, tas, tpose, shiftX, shiftY, f, ppx, ppy, thr, alpha, beta, maxReproj, sub);
}

double ref_esac_backward(float* sc, float* out_grads, int E, int H, int W, int64_t* assign, int N, float* gt_pose16,
                         float wRot, float wTrans, float cut, int shiftX, int shiftY, float f, float ppx, float ppy,
                         float thr, float alpha, float beta, float maxReproj, int sub) {
    at::Tensor tsc(sc, {E, 3, H, W}), tg(out_grads, {E, 3, H, W}), tas(assign, {N}), tgt(gt_pose16, {4, 4});
    return esac_backward(tsc, tg, tas, tgt, wRot, wTrans, cut, shiftX, shiftY, f, ppx, ppy, thr, alpha, beta, maxReproj, sub);
}
}
