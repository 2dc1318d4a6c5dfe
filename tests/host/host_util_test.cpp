// host_util_test.cpp -- the library's host-only code (tracking_sdf_amd/csrc/host_util.{hpp,cpp}: staging thread pool,
// cloud repacking, sample gather, slab arithmetic, the shared-memory rendezvous and fan-in of the ranks of one node,
// host_math.hpp) exercised WITHOUT a device, so that it can run under -fsanitize=address,undefined and -fsanitize=thread.
// Built and run by tests/test_host_sanitized.py (pytest -m "not gpu").  Usage: host_util_test <case> ; exit code 0 = pass.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../tracking_sdf_amd/csrc/host_math.hpp"
#include "../../tracking_sdf_amd/csrc/host_util.hpp"

using namespace tsdf;
using namespace tsdf::host;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

// ---- the staging pool: many short jobs in quick succession, parts disjoint, nothing lost, sleepers woken
static int case_pool() {
    for (int workers : {0, 1, 3, 7}) {
        HostPool pool(workers);
        CHECK(pool.parts() == workers + 1);
        std::vector<long long> acc(64, 0);
        long long expect = 0;
        for (int job = 0; job < 3000; ++job) {
            const int n = 1 + job % 97;
            std::vector<int> hit((size_t)n, 0);
            const std::function<void(int, int)> fn = [&](int part, int parts) {
                const int i0 = n * part / parts, i1 = n * (part + 1) / parts;
                for (int i = i0; i < i1; ++i) { hit[(size_t)i] += 1; acc[(size_t)part] += i; }
            };
            pool.run(fn);
            for (int i = 0; i < n; ++i) { CHECK(hit[(size_t)i] == 1); expect += i; }
            if (job % 500 == 0) usleep(2000);          // let the workers fall asleep: the next run() must wake them
        }
        long long got = 0;
        for (long long a : acc) got += a;
        CHECK(got == expect);
    }
    return 0;
}

// ---- PCL-style clouds -> planes: the 4-points-at-a-time path against a per-point copy, every alignment and layout
static int case_repack() {
    std::mt19937 rng(7);
    struct Lay { int ps, xo, r, g, b, ns, no; };
    const Lay lays[] = {{32, 0, 18, 17, 16, 32, 0}, {16, 0, 12, 13, 14, 16, 0}, {12, 0, -1, -1, -1, 12, 0}, {48, 8, 40, 41, 42, 20, 4}, {15, 1, 13, 14, 13, 13, 1}};
    for (const Lay& l : lays) {
        for (size_t n : {1u, 3u, 4u, 5u, 63u, 64u, 1027u}) {
            tsdf_aos_layout lay{};
            lay.point_stride = l.ps; lay.xyz_offset = l.xo; lay.r_offset = l.r; lay.g_offset = l.g; lay.b_offset = l.b;
            lay.normal_stride = l.ns; lay.normal_offset = l.no;
            const bool color = l.r >= 0;
            // exact-size heap blocks: a read past the last struct is caught by the address sanitizer
            std::vector<unsigned char> pts(n * (size_t)l.ps), nrm(n * (size_t)l.ns);
            for (auto& b : pts) b = (unsigned char)rng();
            for (auto& b : nrm) b = (unsigned char)rng();
            for (size_t off : {0u, 1u, 2u, 3u}) {                           // the plane's start relative to 16 bytes
                std::vector<float> px(3 * n + 8, -1.f), pn(3 * n + 8, -1.f), rx(3 * n + 8, -1.f), rn(3 * n + 8, -1.f);
                std::vector<uint8_t> pc(3 * n + 8, 9), rc(3 * n + 8, 9);
                float* const dx = px.data() + off; float* const dn = pn.data() + off;
                for (size_t i0 : {(size_t)0, n / 3}) {
                    const size_t i1 = n;
                    repack_aos(lay, pts.data(), nrm.data(), color, dx, dn, pc.data(), i0, i1);
                    for (size_t i = i0; i < i1; ++i) {
                        std::memcpy(rx.data() + off + 3 * i, pts.data() + i * l.ps + l.xo, 12);
                        std::memcpy(rn.data() + off + 3 * i, nrm.data() + i * l.ns + l.no, 12);
                        if (color) { rc[3 * i] = pts[i * l.ps + l.r]; rc[3 * i + 1] = pts[i * l.ps + l.g]; rc[3 * i + 2] = pts[i * l.ps + l.b]; }
                    }
                    CHECK(std::memcmp(px.data(), rx.data(), px.size() * 4) == 0);
                    CHECK(std::memcmp(pn.data(), rn.data(), pn.size() * 4) == 0);
                    CHECK(pc == rc);
                    CHECK(points_equal_planes(lay, pts.data(), color, dx, pc.data(), i0, i1));
                    CHECK(normals_equal_plane(lay, nrm.data(), dn, i0, i1));
                }
                if (n > 4) {                                             // one byte of one point changed in place
                    pts[(n / 2) * (size_t)l.ps + (size_t)l.xo + 5] ^= 0x40;
                    CHECK(!points_equal_planes(lay, pts.data(), color, dx, pc.data(), 0, n));
                    pts[(n / 2) * (size_t)l.ps + (size_t)l.xo + 5] ^= 0x40;
                    nrm[(n - 1) * (size_t)l.ns + (size_t)l.no + 11] ^= 1;
                    CHECK(!normals_equal_plane(lay, nrm.data(), dn, 0, n));
                    nrm[(n - 1) * (size_t)l.ns + (size_t)l.no + 11] ^= 1;
                }
            }
        }
    }
    return 0;
}

// ---- the tracker's sample list in the reference's visiting order (camera_tracking.cpp:162-163)
static int case_gather() {
    const int w = 37, h = 23, st = 3, ncols = (w + st - 1) / st, nrows = (h + st - 1) / st;
    std::vector<float> xyz((size_t)w * h * 3);
    for (size_t i = 0; i < xyz.size(); ++i) xyz[i] = (float)i;
    std::vector<float> out((size_t)ncols * nrows * 4, -1.f);
    for (int part = 0; part < 4; ++part) gather_samples(xyz.data(), 12, 0, w, st, ncols, nrows, nrows * part / 4, nrows * (part + 1) / 4, out.data());
    for (int ci = 0; ci < ncols; ++ci)
        for (int rj = 0; rj < nrows; ++rj) {
            const float* o = &out[4 * ((size_t)ci * nrows + rj)];
            const size_t pix = (size_t)(rj * st) * w + (size_t)(ci * st);
            CHECK(o[0] == xyz[3 * pix] && o[1] == xyz[3 * pix + 1] && o[2] == xyz[3 * pix + 2] && o[3] == 0.f);
        }
    return 0;
}

// ---- slabs: every cut is a partition, no rank is empty, weights lower the busiest rank, bad input is refused
static int case_slabs() {
    tsdf_config c;
    tsdf_default_config(&c);
    const double K[9] = {525, 0, 319.5, 0, 525, 239.5, 0, 0, 1}, R0[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0}, t0[3] = {0, 0, 1};
    for (int m : {64, 256, 512}) {
        c.m = m;
        std::vector<double> wts((size_t)m, 0.0);
        CHECK(tsdf_frustum_layer_weights(&c, K, 640, 480, R0, t0, 5.0f, wts.data()) == TSDF_OK);
        for (double v : wts) CHECK(std::isfinite(v) && v > 0.0);
        for (int n : {1, 2, 3, 8}) {
            for (int halo : {0, 6}) {
                std::vector<double> pre((size_t)m + 1, 0.0);
                for (int i = 0; i < m; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + wts[(size_t)i];
                auto cost = [&](int a, int b) { const int lo = a - halo < 0 ? 0 : a - halo, hi = b + halo > m ? m : b + halo; return pre[(size_t)hi] - pre[(size_t)lo]; };
                int prev = 0;
                double worst_w = 0, worst_u = 0;
                for (int r = 0; r < n; ++r) {
                    int32_t x0, x1, u0, u1;
                    CHECK(tsdf_slab_range_weighted(m, n, r, halo, wts.data(), &x0, &x1) == TSDF_OK);
                    CHECK(tsdf_slab_range(m, n, r, &u0, &u1) == TSDF_OK);
                    CHECK(x0 == prev && x1 > x0);
                    prev = x1;
                    worst_w = std::max(worst_w, cost(x0, x1)); worst_u = std::max(worst_u, cost(u0, u1));
                }
                CHECK(prev == m);
                CHECK(worst_w <= worst_u * (1.0 + 1e-9));
            }
        }
        int32_t a, b;
        std::vector<double> bad = wts; bad[3] = std::nan("");
        CHECK(tsdf_slab_range_weighted(m, 2, 0, 0, bad.data(), &a, &b) == TSDF_E_BADARG);
        bad[3] = -1.0;
        CHECK(tsdf_slab_range_weighted(m, 2, 0, 0, bad.data(), &a, &b) == TSDF_E_BADARG);
    }
    // tsdf_frustum_layer_weights refuses what would poison the weights
    c.m = 64;
    std::vector<double> w((size_t)64, 0.0);
    double Kb[9]; std::memcpy(Kb, K, sizeof Kb); Kb[0] = 0.0;
    CHECK(tsdf_frustum_layer_weights(&c, Kb, 640, 480, R0, t0, 5.0f, w.data()) == TSDF_E_BADARG);
    Kb[0] = std::nan("");
    CHECK(tsdf_frustum_layer_weights(&c, Kb, 640, 480, R0, t0, 5.0f, w.data()) == TSDF_E_BADARG);
    tsdf_config cz = c; cz.width = 0.f;
    CHECK(tsdf_frustum_layer_weights(&cz, K, 640, 480, R0, t0, 5.0f, w.data()) == TSDF_E_BADARG);
    const double tb[3] = {0, INFINITY, 1};
    CHECK(tsdf_frustum_layer_weights(&c, K, 640, 480, R0, tb, 5.0f, w.data()) == TSDF_E_BADARG);
    for (double v : w) CHECK(v == 0.0);                                   // nothing was added by the refused calls
    CHECK(tsdf_halo_for(&c, 6.0f) > 0);
    return 0;
}

// ---- host_math: the Gauss-Newton step on a hand-made system (KAT-4 / KAT-6 of SURVEY 8c through the C ABI)
static int case_math() {
    double rot[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0}, trans[3] = {0, 0, 1}, ri[9], rit[3];
    CHECK(tsdf_host_set_pose(rot, trans, ri, rit) == TSDF_OK);
    for (int a = 0; a < 3; ++a)                                           // rot_inv * rot = I
        for (int b = 0; b < 3; ++b) {
            double v = 0; for (int k = 0; k < 3; ++k) v += ri[3 * a + k] * rot[3 * k + b];
            CHECK(std::fabs(v - (a == b ? 1.0 : 0.0)) < 1e-15);
        }
    double A[36] = {0}, b[6] = {1, 0, 0, 0, 0, 0}, tw[6];
    for (int a = 0; a < 6; ++a) A[7 * a] = 1.0;
    int32_t stop = -1;
    CHECK(tsdf_host_gn_step(rot, trans, A, b, 0.001f, tw, &stop) == TSDF_OK);
    CHECK(tw[0] == 1.0 && stop == 0);
    double As[36] = {0};                                                  // singular: refused, pose untouched
    double r2[9], t2[3]; std::memcpy(r2, rot, sizeof r2); std::memcpy(t2, trans, sizeof t2);
    CHECK(tsdf_host_gn_step(r2, t2, As, b, 0.001f, tw, &stop) == TSDF_E_SINGULAR);
    CHECK(std::memcmp(r2, rot, sizeof r2) == 0 && std::memcmp(t2, trans, sizeof t2) == 0);
    double rpm[54];
    CHECK(tsdf_host_perturbed_rotations(rot, 0.01f, rpm) == TSDF_OK);
    for (double v : rpm) CHECK(std::isfinite(v));
    return 0;
}

// ---- the shared-memory rendezvous and fan-in with N rank PROCESSES (the parent only forks and waits)
static int rank_main(const char* name, int nranks, int rank, int passes) {
    ShmSegment seg;
    std::string err;
    const int rc = shm_rendezvous(name, nranks, rank, &seg, &err);
    if (rc) { std::fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 10; }
    for (int seq = 1; seq <= passes; ++seq) {
        double row[kShmRowDoubles], red[kShmRowDoubles];
        for (int e = 0; e < kShmRowDoubles; ++e) row[e] = (double)(rank + 1) * (e + 1) + seq;
        shm_publish(seg, (unsigned long long)seq, row);
        if (shm_fan_in(seg, (unsigned long long)seq, 30, red, &err)) { std::fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 11; }
        for (int e = 0; e < kShmRowDoubles; ++e) {
            double want = row[e];                                         // beyond the summed part: this rank's own
            if (e < 30) { want = 0; for (int r = 0; r < nranks; ++r) want += (double)(r + 1) * (e + 1) + seq; }
            if (red[e] != want) { std::fprintf(stderr, "rank %d pass %d entry %d: %g != %g\n", rank, seq, e, red[e], want); return 12; }
        }
    }
    shm_unmap(&seg);
    return 0;
}
static int case_shm(int nranks) {
    char name[64];
    std::snprintf(name, sizeof name, "/tsdf_host_test_%d", (int)getpid());
    for (int round = 0; round < 2; ++round) {
        if (round == 1) {
            // a leftover of a "crashed job" under the same name (another size, never completed): rank 0 must replace it and
            // the others must not mistake it for theirs
            ShmSegment junk; std::string e;
            const pid_t p = fork();
            if (p == 0) { alarm(3); shm_rendezvous(name, nranks + 3, 0, &junk, &e); _exit(0); }   // creates the segment, then waits in vain
            usleep(200000);
            kill(p, SIGKILL); int st; waitpid(p, &st, 0);
        }
        std::vector<pid_t> kids;
        for (int r = 0; r < nranks; ++r) {
            const pid_t p = fork();
            if (p < 0) return 1;
            if (p == 0) { alarm(60); _exit(rank_main(name, nranks, r, 200)); }
            kids.push_back(p);
            if (r == 0 && round == 0) usleep(1000);
        }
        int bad = 0;
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
        CHECK(bad == 0);
    }
    return 0;
}

int main(int argc, char** argv) {
    const std::string c = argc > 1 ? argv[1] : "";
    if (c == "pool") return case_pool();
    if (c == "repack") return case_repack();
    if (c == "gather") return case_gather();
    if (c == "slabs") return case_slabs();
    if (c == "math") return case_math();
    if (c == "shm") return case_shm(argc > 2 ? std::atoi(argv[2]) : 2);
    std::fprintf(stderr, "usage: host_util_test pool|repack|gather|slabs|math|shm [ranks]\n");
    return 2;
}
