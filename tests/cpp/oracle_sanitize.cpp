// -fsanitize=address,undefined build of the CPU oracle (test infrastructure), driven over every stage of the path on a small synthetic pair
// in all three arithmetic modes, with the edge inputs the parity tests use (empty, one point, duplicates, NaN / inf rows).  Built and run by
// tests/test_oracle_sanitize.py in the CPU suite: the oracle is what every parity claim is checked against, so it gets the
// sanitizers the GPU box cannot offer.  The translation units are compiled INTO this program (no shared library, no LD_PRELOAD).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../oracle/lgr_oracle.h"

static std::vector<float> cloud(int n, unsigned seed, float shift) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> N(0.f, 0.003f);
    std::vector<float> p((size_t) n * 12, 0.f);
    for (int i = 0; i < n; ++i) {
        float x = 3.f * U(g) + shift, y = 2.f * U(g);
        float z = 0.25f * std::sin(2.1f * x + 0.4f * y) + 0.2f * std::cos(1.3f * y - 0.7f * x) + (x > 1.5f + shift ? 0.4f : 0.f);
        float* q = &p[(size_t) i * 12];
        q[0] = x + N(g); q[1] = y + N(g); q[2] = z + N(g); q[3] = 1.f; q[8] = 1.f;
    }
    return p;
}

int main() {
    const int n = 6000;
    std::vector<float> src = cloud(n, 566, 0.f), tgt = cloud(n, 567, 0.6f);
    orc_set_num_threads(4);
    int bad = 0;
    for (int mode : {(int) ORC_ARITH_ROUND4, (int) ORC_ARITH_CANONICAL, (int) ORC_ARITH_PCL}) {
        orc_set_arith_mode(mode);
        std::vector<float> ds((size_t) n * 12), dt((size_t) n * 12);
        int ns = 0, nt = 0;
        bad += orc_downsample(src.data(), n, 0.05f, 1, ds.data(), &ns) != 0;
        bad += orc_downsample(tgt.data(), n, 0.05f, 0, dt.data(), &nt) != 0;
        const float vp[3] = {0.f, 0.f, 10.f};
        bad += orc_normals_knn(ds.data(), ns, nullptr, 0, 30, vp, 0) != 0;
        bad += orc_normals_knn(dt.data(), nt, nullptr, 0, 30, vp, 0) != 0;
        ds[4] = std::numeric_limits<float>::quiet_NaN();                      // a NaN normal
        std::vector<float> kp = src;
        kp[12 * 7] = std::numeric_limits<float>::infinity();                  // an invalid key point
        kp[12 * 9] = 1e4f;                                                     // a key point without neighbours
        std::vector<float> fs((size_t) n * 33), ft((size_t) n * 33), sp((size_t) ns * 33);
        bad += orc_fpfh(kp.data(), n, ds.data(), ns, 0.25f, fs.data(), 0) != 0;
        bad += orc_fpfh(tgt.data(), n, dt.data(), nt, 0.25f, ft.data(), 0) != 0;
        bad += orc_spfh(ds.data(), ns, 0.25f, sp.data(), 0) != 0;
        std::vector<int> ij(n), ji(n);
        std::vector<float> dij(n), dji(n);
        bad += orc_match_bf(fs.data(), n, ft.data(), n, 1000, ij.data(), dij.data()) != 0;
        bad += orc_match_bf(ft.data(), n, fs.data(), n, 1000, ji.data(), dji.data()) != 0;
        bad += ij[7] != -1;                                                    // NaN rows never match
        std::vector<float> den(n);
        bad += orc_smoothed_densities(src.data(), n, 2, den.data()) != 0;
        lgr_orc_params p;
        orc_default_params(&p);
        p.feature_radius = 0.25f; p.distance_thr = 0.1f; p.bf_block_size = 1000; p.max_iterations = 20000; p.rng_mode = ORC_RNG_PHILOX;
        p.has_vp_src = p.has_vp_tgt = 1; p.vp_src[2] = p.vp_tgt[2] = 10.f;
        for (int matching : {(int) ORC_MATCH_LR, (int) ORC_MATCH_CLUSTER, (int) ORC_MATCH_ONE_SIDED}) {
            p.matching_id = matching;
            lgr_orc_result res;
            std::vector<lgr_orc_corr> corr(n);
            int nc = 0;
            bad += orc_align(src.data(), n, tgt.data(), n, &p, &res, corr.data(), &nc, nullptr) != 0;
            if (matching == ORC_MATCH_LR && nc >= 30) {
                float T[16];
                int diag[8];
                float ang = 0.f;
                bad += orc_gror(src.data(), n, tgt.data(), n, corr.data(), nc, 0.1f, 800, T, diag, &ang) != 0;
                for (int metric : {(int) ORC_METRIC_CORRESPONDENCES, (int) ORC_METRIC_CLOSEST_PLANE, (int) ORC_METRIC_COMBINATION}) {
                    lgr_orc_params q = p;
                    q.metric_id = metric; q.max_iterations = 4000;
                    std::vector<unsigned char> mask(nc);
                    bad += orc_ransac(src.data(), n, tgt.data(), n, corr.data(), nc, &q, &res, mask.data()) != 0;
                }
            }
        }
        // edge inputs: empty, a single point, exact duplicates
        int n0 = -1;
        bad += orc_downsample(src.data(), 0, 0.05f, 1, ds.data(), &n0) != 0 || n0 != 0;
        bad += orc_downsample(src.data(), 1, 0.05f, 0, ds.data(), &n0) != 0 || n0 != 1;
        std::vector<float> dup((size_t) 8 * 12);
        for (int i = 0; i < 8; ++i) std::memcpy(&dup[(size_t) i * 12], src.data(), 48);
        bad += orc_normals_knn(dup.data(), 8, nullptr, 0, 30, vp, 0) != 0;
        std::vector<float> fd(8 * 33);
        bad += orc_fpfh(dup.data(), 8, dup.data(), 8, 0.25f, fd.data(), 0) != 0;
    }
    orc_set_arith_mode(ORC_ARITH_CANONICAL);
    // the libm restatement on its special values
    const float sp[] = {0.f, -0.f, 1.f, -1.f, 0.5f, 2.f, INFINITY, -INFINITY, NAN, 1e-38f, 1e-45f, 3e38f};
    float out[12];
    for (int fn = 0; fn < 5; ++fn) {
        if (fn >= 3) { const float sc[] = {0.f, 0.3f, 0.78f, 0.8f, 1.04f, 2.f, 3.f, 6.f, 1e-5f, -1.f, -3.f, 100.f}; bad += orc_libm_eval(fn, sc, sc, 12, out) != 0; }
        else bad += orc_libm_eval(fn, sp, sp, 12, out) != 0;
    }
    unsigned ctr[4] = {0u, 0u, 0u, 0u}, w[4];
    orc_philox_full(0ull, ctr, w);
    bad += w[0] != 0x6627e8d5u;
    std::printf("oracle_sanitize: %d failures\n", bad);
    return bad ? 1 : 0;
}
