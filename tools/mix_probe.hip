// What does the memory system give for integrate_kernel's access mix with NO arithmetic at all?  (round 3 diagnosis)
// Per item (as the plant scene at 512^3, profiles/r03_*): two paired record gathers (32 records of 32 bytes each, the
// records of a run of pixels 1..2 apart in an L2-resident 9.8 MB table), an 8-byte {D,W} load and a 16-byte colour
// load of a contiguous run of ~34 of the 64 voxels of a random 64-voxel segment of 1 GiB / 2 GiB arrays, and the two
// stores.  198.8k items over 1280 workgroups x 4 wavefronts, like the kernel.  Modes: 0 = everything, 1 = no gathers,
// 2 = no volume traffic, 3 = gathers + loads (no stores).  PIPE = 1 prefetches the next item's gathers; PIPE = 2 pulls the
// NEXT item's volume lines (4 x 128 B of {D,W}, 8 of colour) into L2 with scalar loads one item ahead: does a vector
// load that hits L2 instead of HBM free the CU's vector-memory pipeline sooner?
// Round 4 additions: DENSE = 1 moves the SAME volume bytes as dense 64-lane batches -- every other
// item issues one {D,W} + colour load / store pair whose lanes 0..31 cover 32 contiguous voxels of one segment and lanes
// 32..63 32 contiguous voxels of the segment of the item before (what a wave-private queue of live lanes pops) -- with
// the two gathers per item unchanged.  order = 1 replaces the random segments by the kernel's real address order: the
// band-sorted list hands a workgroup consecutive rows (consecutive 64-voxel chunks of a row, then the next row 4 KiB
// further), dealt item by item to its four wavefronts, and the gathers of a workgroup fall into one moving image column.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE, int PIPE>
__global__ __launch_bounds__(256) void mix(const u4* __restrict__ rec, unsigned nrec, u2* __restrict__ dw, u4* __restrict__ col,
                                           unsigned nseg, int items_per_wave, unsigned* __restrict__ out, unsigned window = 0, unsigned share = 1) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned seed = wave * 2654435761u + 12345u, acc = 0;
    auto gather_addr = [&](unsigned base, unsigned half_of) {      // record of lane pair, 1..2 records apart
        const unsigned pair = half_of * 32 + (lane >> 1);
        return (size_t)((base + pair + (pair >> 1)) % nrec) * 2 + (lane & 1);
    };
    // window > 0: the gathers of `share` consecutive workgroups (share = 1: the four wavefronts of one workgroup) fall into a
    // window of `window` records (480 = one image column) that moves every 8 items -- do concurrent wavefronts that
    // share pixels get them from the L1?
    unsigned wseed = (blockIdx.x / share) * 747796405u + 2891336453u;
    auto pick = [&](int it) -> unsigned {
        if (window == 0) return rnd(seed) % nrec;
        if ((it & 7) == 0) wseed = wseed * 1664525u + 1013904223u;
        return ((wseed >> 8) % (nrec - window) + rnd(seed) % window) % nrec;
    };
    unsigned base = pick(0);
    u4 ga = u4{0, 0, 0, 0}, gb = ga;
    if (MODE != 1 && PIPE == 1) { ga = rec[gather_addr(base, 0)]; gb = rec[gather_addr(base, 1)]; }
    unsigned seg_next = __builtin_amdgcn_readfirstlane(rnd(seed) % nseg);
    unsigned sink = 0;
    u2 pd = u2{0, 0}; u4 pc = u4{0, 0, 0, 0}; unsigned pseg = 0, pfirst = 99u;       // PIPE 3: the item whose volume data is in flight
    for (int it = 0; it < items_per_wave; ++it) {
        if (MODE != 1 && MODE != 4 && PIPE != 1) { ga = rec[gather_addr(base, 0)]; gb = rec[gather_addr(base, 1)]; }
        const unsigned seg = seg_next;
        seg_next = __builtin_amdgcn_readfirstlane(rnd(seed) % nseg);
        if (PIPE == 2 && MODE != 2) {
            // one dword of each 128-byte line of the next item's segments, through the scalar cache
            const char* pd = reinterpret_cast<const char*>(dw) + (size_t)seg_next * 512;
            const char* pc = reinterpret_cast<const char*>(col) + (size_t)seg_next * 1024;
            unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;
            asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                         "s_load_dword %0, %12, 0x0\n\ts_load_dword %1, %12, 0x80\n\ts_load_dword %2, %12, 0x100\n\ts_load_dword %3, %12, 0x180\n\t"
                         "s_load_dword %4, %13, 0x0\n\ts_load_dword %5, %13, 0x80\n\ts_load_dword %6, %13, 0x100\n\ts_load_dword %7, %13, 0x180\n\t"
                         "s_load_dword %8, %13, 0x200\n\ts_load_dword %9, %13, 0x280\n\ts_load_dword %10, %13, 0x300\n\ts_load_dword %11, %13, 0x380\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10), "=&s"(t11)
                         : "s"(pd), "s"(pc) : "memory");
            sink ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7 ^ t8 ^ t9 ^ t10 ^ t11;
        }
        const unsigned first = __builtin_amdgcn_readfirstlane(rnd(seed) % 30u);
        const bool live = lane >= first && lane < first + 34u;
        acc ^= ga.x + gb.w;                                        // consume the gathers
        const unsigned nbase = pick(it + 1);
        u4 na = u4{0, 0, 0, 0}, nb = na;
        if (MODE != 1 && PIPE == 1 && it + 1 < items_per_wave) { na = rec[gather_addr(nbase, 0)]; nb = rec[gather_addr(nbase, 1)]; }
        if (MODE != 2 && PIPE != 3) {
            u2 d = u2{0, 0}; u4 c = u4{0, 0, 0, 0};
            if (live) { d = dw[(size_t)seg * 64 + lane]; c = __builtin_nontemporal_load(&col[(size_t)seg * 64 + lane]); }
            d.x += acc; d.y += acc ^ 1u; c.x += d.x; c.y ^= d.y; c.z += acc; c.w ^= acc + 3u;   /* every component changes: round 4's form let hipcc drop the unchanged ones from loads and stores */
            // MODE 4 (no gathers): the stores go to ANOTHER random segment (copy-like)
            const unsigned wseg = MODE == 4 ? __builtin_amdgcn_readfirstlane((seg * 2654435761u) % nseg) : seg;
            if (MODE != 3 && live) { dw[(size_t)wseg * 64 + lane] = d; __builtin_nontemporal_store(c, &col[(size_t)wseg * 64 + lane]); }
            if (MODE == 3) acc ^= d.x + c.y;
        }
        if (MODE != 2 && PIPE == 3) {
            // volume loads one item ahead: the loads of item it + 1 are in flight while item it is stored
            const unsigned nfirst = (seg_next * 40503u >> 7) % 30u;
            const bool nlive = lane >= nfirst && lane < nfirst + 34u;
            u2 nd = u2{0, 0}; u4 nc = u4{0, 0, 0, 0};
            if (it + 1 < items_per_wave && nlive) { nd = dw[(size_t)seg_next * 64 + lane]; nc = __builtin_nontemporal_load(&col[(size_t)seg_next * 64 + lane]); }
            const bool plive = pfirst != 99u && lane >= pfirst && lane < pfirst + 34u;
            u2 d = pd; u4 c = pc;
            d.x += acc; d.y += acc ^ 1u; c.x += d.x; c.y ^= d.y; c.z += acc; c.w ^= acc + 3u;   /* every component changes: round 4's form let hipcc drop the unchanged ones from loads and stores */
            if (MODE != 3 && plive) { dw[(size_t)pseg * 64 + lane] = d; __builtin_nontemporal_store(c, &col[(size_t)pseg * 64 + lane]); }
            if (MODE == 3) acc ^= d.x + c.y;
            pd = nd; pc = nc; pseg = seg_next; pfirst = it + 1 < items_per_wave ? nfirst : 99u;
        }
        base = nbase; ga = na; gb = nb;
    }
    if ((acc ^ sink) == 0x12345678u) out[wave] = acc;
}

// DENSE / ordered variant (see the header comment).  live lanes per item: 34 of 64 (a run at a random start) when
// DENSE = 0; DENSE = 1: batches of 64 = 2 x 32 contiguous voxels, one batch per two items (=> 32 per item: same bytes +-6 %).
template <int MODE, int DENSE>
__global__ __launch_bounds__(256) void mix2(const u4* __restrict__ rec, unsigned nrec, u2* __restrict__ dw, u4* __restrict__ col,
                                            unsigned nseg, int items_per_wave, unsigned* __restrict__ out, int order) {
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 4 + wv;
    unsigned seed = wave * 2654435761u + 12345u, acc = 0;
    // ordered: the workgroup's share of the list starts at a row of its own; items wg_first + wv + 4 it; a row has 8 chunks
    // of which ~5 are listed (the frustum interval), i.e. list index -> segment = (idx / 5) * 8 + 1 + idx % 5
    const unsigned wg_first = (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % (nseg / 8u - (unsigned)items_per_wave)) * 8u;
    auto seg_of = [&](int it) -> unsigned {
        if (!order) return rnd(seed) % nseg;
        const unsigned idx = (unsigned)it * 4u + wv;
        return (wg_first + (idx / 5u) * 8u + 1u + idx % 5u) % nseg;
    };
    // ordered gathers: one image column (480 records) per workgroup, moving on every 8 items; the lanes' records 1..2 apart
    unsigned wseed = blockIdx.x * 747796405u + 2891336453u;
    auto pick = [&](int it) -> unsigned {
        if (!order) return rnd(seed) % nrec;
        if ((it & 7) == 0) wseed = wseed * 1664525u + 1013904223u;
        return ((wseed >> 8) % (nrec - 480u) + rnd(seed) % 384u) % nrec;
    };
    auto gather_addr = [&](unsigned base, unsigned half_of) {
        const unsigned pair = half_of * 32 + (lane >> 1);
        return (size_t)((base + pair + (pair >> 1)) % nrec) * 2 + (lane & 1);
    };
    unsigned prev_seg = 0;
    for (int it = 0; it < items_per_wave; ++it) {
        u4 ga = u4{0, 0, 0, 0}, gb = ga;
        if (MODE != 1) { const unsigned base = pick(it); ga = rec[gather_addr(base, 0)]; gb = rec[gather_addr(base, 1)]; }
        const unsigned seg = __builtin_amdgcn_readfirstlane(seg_of(it));
        const unsigned first = __builtin_amdgcn_readfirstlane(rnd(seed) % 30u);
        acc ^= ga.x + gb.w;
        if (MODE != 2) {
            size_t vox; bool live;
            if (DENSE) {
                live = (it & 1) != 0;                                         // a batch every other item, all 64 lanes
                vox = lane < 32 ? (size_t)seg * 64 + first + lane : (size_t)prev_seg * 64 + (first ^ 21u) % 30u + (lane - 32);
            } else {
                live = lane >= first && lane < first + 34u;
                vox = (size_t)seg * 64 + lane;
            }
            u2 d = u2{0, 0}; u4 c = u4{0, 0, 0, 0};
            if (live) { d = dw[vox]; c = __builtin_nontemporal_load(&col[vox]); }
            d.x += acc; d.y += acc ^ 1u; c.x += d.x; c.y ^= d.y; c.z += acc; c.w ^= acc + 3u;   /* every component changes: round 4's form let hipcc drop the unchanged ones from loads and stores */
            if (MODE != 3 && live) { dw[vox] = d; __builtin_nontemporal_store(c, &col[vox]); }
            if (MODE == 3) acc ^= d.x + c.y;
        }
        prev_seg = seg;
    }
    if (acc == 0x12345678u) out[wave] = acc;
}

// TILE pattern (round 4): what if a wavefront's 64 lanes were 16 consecutive rows (j: the viewing direction for the
// reference's initial pose, 4 KiB apart in {D,W}) x 4 consecutive k instead of 64 consecutive k of one row?  The four
// wavefronts of a workgroup take the four k-quads of a 16 x 16 tile (together whole 128-byte lines of 16 rows), walk the
// four tiles of a 64-voxel chunk, then the next chunk, then the next 16 rows.  The 16 rows of a lane group see nearly the
// same pixels (a ray through the volume), so a gather instruction touches a few cache lines instead of ~39.
// Same number of lanes, same live fraction (a run of 34 of the 64 k of every row and chunk), same bytes.
template <int MODE>
__global__ __launch_bounds__(256) void mix3(const u4* __restrict__ rec, unsigned nrec, u2* __restrict__ dw, u4* __restrict__ col,
                                            unsigned nrows, int steps, unsigned* __restrict__ out, int drift16 /* pixel columns crossed by 16 rows, x16 */) {
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 4 + wv;
    const unsigned jj = lane >> 2, kk = lane & 3;
    unsigned acc = 0;
    const unsigned row0 = (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % (nrows - 16u * ((unsigned)steps / 20u + 2u))) & ~15u;
    const unsigned colbase = (blockIdx.x * 747796405u + 2891336453u) % 600u;
    for (int it = 0; it < steps; ++it) {
        const unsigned tk = (unsigned)it & 3u, grp = (unsigned)it >> 2, chunk = grp % 5u + 1u, rowblk = grp / 5u;
        const unsigned row = row0 + rowblk * 16u + jj;
        const unsigned k = chunk * 64u + tk * 16u + wv * 4u + kk;
        unsigned h = (blockIdx.x * 2654435761u) ^ (grp * 40503u + jj * 7u);   // the row's live run in this chunk
        h = (h ^ (h >> 13)) * 1274126177u;
        const unsigned first = (h >> 8) % 30u;
        const bool live = (k & 63u) >= first && (k & 63u) < first + 34u;
        u4 ga = u4{0, 0, 0, 0}, gb = ga;
        if (MODE != 1) {
            // record of lane pair p: pixel column drifts with the row (jj), pixel row = 2.4 k; column-major records
            auto recidx = [&](unsigned p) {
                const unsigned pj = p >> 2, pk = chunk * 64u + tk * 16u + wv * 4u + (p & 3u);
                const unsigned colpx = colbase + rowblk + ((pj * (unsigned)drift16) >> 8), rowpx = (pk * 12u / 5u) % 480u;
                return (colpx * 480u + rowpx) % nrec;
            };
            ga = rec[(size_t)recidx(lane >> 1) * 2 + (lane & 1)];
            gb = rec[(size_t)recidx(32u + (lane >> 1)) * 2 + (lane & 1)];
        }
        acc ^= ga.x + gb.w;
        if (MODE != 2) {
            const size_t vox = (size_t)row * 512u + k;
            u2 d = u2{0, 0}; u4 c = u4{0, 0, 0, 0};
            if (live) { d = dw[vox]; c = __builtin_nontemporal_load(&col[vox]); }
            d.x += acc; d.y += acc ^ 1u; c.x += d.x; c.y ^= d.y; c.z += acc; c.w ^= acc + 3u;   /* every component changes: round 4's form let hipcc drop the unchanged ones from loads and stores */
            if (live) { dw[vox] = d; __builtin_nontemporal_store(c, &col[vox]); }
        }
    }
    if (acc == 0x12345678u) out[wave] = acc;
}


// SPAN pattern (round 6, VERDICT r5 item 2a / north_star's "depth tiles staged in LDS"): the 64 records an item gathers lie in
// ONE contiguous span of the column-major record table (the pixels a 64-voxel k-run projects to: ~96 records = 3 KiB at
// 1.5 records per voxel).  SRC 0 = what integrate_kernel does today: two paired 64-address gathers (lane l: half l&1 of the
// record of lane l>>1 / 32 + l>>1) + the un-shuffle through a wave-private LDS buffer.  SRC 1 = the span read COALESCED
// (SPAN_KB x 1 KiB wave loads, lane-contiguous 16 bytes each: no 64-address work in the texture addresser), written to the
// wave-private LDS buffer, each lane then reads its own record with two ds_read_b128.  SRC 2 = the same span by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  Same address order, same volume traffic, gathers requested
// one item ahead in all three (as the kernel's software pipeline does).
template <int MODE, int SRC, int SPAN_KB>
__global__ __launch_bounds__(256) void mix4(const u4* __restrict__ rec, unsigned nrec, u2* __restrict__ dw, u4* __restrict__ col,
                                            unsigned nseg, int items_per_wave, unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 4 + wv;
    unsigned seed = wave * 2654435761u + 12345u, acc = 0;
    const unsigned wg_first = (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % (nseg / 8u - (unsigned)items_per_wave)) * 8u;
    auto seg_of = [&](int it) -> unsigned {
        const unsigned idx = (unsigned)it * 4u + wv;
        return (wg_first + (idx / 5u) * 8u + 1u + idx % 5u) % nseg;
    };
    unsigned wseed = blockIdx.x * 747796405u + 2891336453u;
    auto pick = [&](int it) -> unsigned {
        if ((it & 7) == 0) wseed = wseed * 1664525u + 1013904223u;
        return ((wseed >> 8) % (nrec - 480u) + rnd(seed) % 384u) % (nrec - 64u * SPAN_KB);
    };
    // wave-private staging: two buffers of SPAN_KB KiB (the span of item it+1 lands while item it is consumed)
    __shared__ u4 s_span[4][2][64 * SPAN_KB];
    const unsigned own = lane + (lane >> 1);                       // own record inside the span (1.5 records per voxel)
    u4 ga = u4{0, 0, 0, 0}, gb = ga, gc = ga, gd = ga;
    auto request = [&](unsigned base, int buf) {
        if (SRC == 0) {
            const unsigned pa = (lane >> 1), pb = 32u + (lane >> 1);
            ga = rec[(size_t)(base + pa + (pa >> 1)) * 2 + (lane & 1)];
            gb = rec[(size_t)(base + pb + (pb >> 1)) * 2 + (lane & 1)];
        } else if (SRC == 1) {
            const u4* src = rec + (size_t)base * 2 + lane;
            ga = src[0]; gb = src[64]; gc = src[128];
            if (SPAN_KB > 3) gd = src[192];
        } else {
            const u4* src = rec + (size_t)base * 2 + lane;
#pragma unroll
            for (int q = 0; q < SPAN_KB; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * q),
                                                 (__attribute__((address_space(3))) void*)(&s_span[wv][buf][64 * q]), 16, 0, 0);
        }
    };
    auto consume = [&](int buf) {
        u4* st = s_span[wv][buf];
        u4 P, N;
        if (SRC == 0) {
            st[lane] = ga; st[64 + lane] = gb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            P = st[2 * lane]; N = st[2 * lane + 1];
        } else {
            if (SRC == 1) {
                st[lane] = ga; st[64 + lane] = gb; st[128 + lane] = gc;
                if (SPAN_KB > 3) st[192 + lane] = gd;
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the probe has no other vector-memory operation in flight here)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            P = st[2 * own]; N = st[2 * own + 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        return (P.x + P.y) ^ (P.z + P.w) ^ (N.x + N.y) ^ (N.z + N.w);          // all 32 bytes of the record are used
    };
    if (MODE != 1) request(pick(0), 0);
    for (int it = 0; it < items_per_wave; ++it) {
        const unsigned seg = __builtin_amdgcn_readfirstlane(seg_of(it));
        const unsigned first = __builtin_amdgcn_readfirstlane(rnd(seed) % 30u);
        if (MODE != 1) {
            if (SRC == 2) {
                // LDS-DMA lands in the buffer directly: item it's span must be complete before item it+1's is requested into the
                // other buffer only in the sense of the counter -- both are vmcnt operations in order, so wait for the older one
                acc ^= consume(it & 1);
                if (it + 1 < items_per_wave) request(pick(it + 1), (it + 1) & 1);
            } else {
                const u4 a = ga, b = gb, c = gc, d = gd;
                u4 na = a, nb = b, nc = c, nd = d;
                if (it + 1 < items_per_wave) { request(pick(it + 1), (it + 1) & 1); na = ga; nb = gb; nc = gc; nd = gd; }
                ga = a; gb = b; gc = c; gd = d;
                acc ^= consume(it & 1);
                ga = na; gb = nb; gc = nc; gd = nd;
            }
        }
        if (MODE != 2) {
            const bool live = lane >= first && lane < first + 34u;
            const size_t vox = (size_t)seg * 64 + lane;
            u2 d = u2{0, 0}; u4 c = u4{0, 0, 0, 0};
            if (live) { d = dw[vox]; c = __builtin_nontemporal_load(&col[vox]); }
            d.x += acc; d.y += acc ^ 1u; c.x += d.x; c.y ^= d.y; c.z += acc; c.w ^= acc + 3u;   /* every component changes: round 4's form let hipcc drop the unchanged ones from loads and stores */
            if (live) { dw[vox] = d; __builtin_nontemporal_store(c, &col[vox]); }
        }
    }
    if (acc == 0x12345678u) out[wave] = acc;
}

int main(int argc, char** argv) {
    const unsigned nrec = 307200, nseg = 2097152;                  // 512^3 / 64 segments
    u4 *rec, *col; u2* dw; unsigned* out;
    CHECK(hipMalloc(&rec, (size_t)nrec * 32)); CHECK(hipMalloc(&dw, (size_t)nseg * 512)); CHECK(hipMalloc(&col, (size_t)nseg * 1024));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(rec, 1, (size_t)nrec * 32)); CHECK(hipMemset(dw, 0, (size_t)nseg * 512)); CHECK(hipMemset(col, 0, (size_t)nseg * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char* names[5] = {"all", "no_gathers", "no_volume", "no_stores", "no_gathers_stores_elsewhere"};
    const bool only_span = argc > 1 && argv[1][0] == 's';      // "span": only round 6's span-through-LDS probe
    const bool only_new = argc > 1;     // any other argument: only the round-4 sweeps
    // grid sweep (199.7k items in all): wavefronts per CU = blocks * 4 / 256
    const int grids[6] = {1280, 256, 512, 768, 1024, 2048};
    for (int gi = 0; gi < (only_new ? 0 : 6); ++gi) {
        const int blocks = grids[gi], ipw = (199680 + blocks * 4 - 1) / (blocks * 4);
        for (int rep = 0; rep < 3; ++rep)
            for (int pipe = 0; pipe < (gi == 0 ? 4 : 1); ++pipe)
                for (int mode = 0; mode < (gi == 0 ? (pipe == 0 ? 5 : 4) : 2); ++mode) {
                    CHECK(hipEventRecord(a));
                    for (int k = 0; k < 10; ++k) {
#define L(M, P) mix<M, P><<<blocks, 256>>>(rec, nrec, dw, col, nseg, ipw, out)
                        if (pipe == 0) { if (mode == 0) L(0, 0); if (mode == 1) L(1, 0); if (mode == 2) L(2, 0); if (mode == 3) L(3, 0); if (mode == 4) L(4, 0); }
                        else if (pipe == 1) { if (mode == 0) L(0, 1); if (mode == 1) L(1, 1); if (mode == 2) L(2, 1); if (mode == 3) L(3, 1); }
                        else if (pipe == 2) { if (mode == 0) L(0, 2); if (mode == 1) L(1, 2); if (mode == 2) L(2, 2); if (mode == 3) L(3, 2); }
                        else { if (mode == 0) L(0, 3); if (mode == 1) L(1, 3); if (mode == 2) L(2, 3); if (mode == 3) L(3, 3); }
                    }
                    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                    if (rep == 2) printf("{\"workgroups\": %d, \"items_per_wavefront\": %d, \"mode\": \"%s\", \"pipe\": %d, \"us_per_launch\": %.1f}\n", blocks, ipw, names[mode], pipe, ms * 100.0);
                }
    }
    for (int wi = 0; wi < (only_new ? 0 : 5); ++wi) {
        const unsigned windows[5] = {0, 480, 960, 480, 480}, shares[5] = {1, 1, 1, 8, 256};
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(a));
            for (int k = 0; k < 10; ++k) mix<0, 0><<<1280, 256>>>(rec, nrec, dw, col, nseg, 39, out, windows[wi], shares[wi]);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("{\"gather_window_records\": %u, \"workgroups_sharing_a_window\": %u, \"mode\": \"all\", \"us_per_launch\": %.1f}\n", windows[wi], shares[wi], ms * 100.0);
        }
    }
    // round 6: the item's pixel records as ONE contiguous span through LDS instead of two 64-address gathers
    for (int span = 3; span <= 4; ++span)
        for (int src = 0; src < 3; ++src)
            for (int mode = 0; mode < 3; ++mode) {
                if (mode == 1 && (src != 0 || span != 3)) continue;          // no gathers: the same launch whatever the source
                const int blocks = 1280, ipw = (199680 + blocks * 4 - 1) / (blocks * 4);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(a));
                    for (int k = 0; k < 10; ++k) {
#define L4(M, S, K) mix4<M, S, K><<<blocks, 256>>>(rec, nrec, dw, col, nseg, ipw, out)
#define L4S(M, S) do { if (span == 3) L4(M, S, 3); else L4(M, S, 4); } while (0)
                        if (src == 0) { if (mode == 0) L4S(0, 0); if (mode == 1) L4S(1, 0); if (mode == 2) L4S(2, 0); }
                        if (src == 1) { if (mode == 0) L4S(0, 1); if (mode == 2) L4S(2, 1); }
                        if (src == 2) { if (mode == 0) L4S(0, 2); if (mode == 2) L4S(2, 2); }
                    }
                    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                    if (rep > 0 && ms < best) best = ms;
                }
                const char* srcs[3] = {"two paired 64-address gathers + LDS un-shuffle (today)", "contiguous span, coalesced loads -> ds_write -> ds_read",
                                       "contiguous span by LDS-DMA (global_load_lds_dwordx4) -> ds_read"};
                printf("{\"probe\": \"mix4\", \"records_from\": \"%s\", \"span_KiB\": %d, \"mode\": \"%s\", \"workgroups\": %d, \"us_per_launch\": %.1f}\n",
                       srcs[src], span, names[mode], blocks, best * 100.0);
            }
    if (only_span) return 0;
    // round 4: dense batches and the kernel's address order
    for (int order = 0; order < 2; ++order)
        for (int dense = 0; dense < 2; ++dense)
            for (int mode = 0; mode < 3; ++mode)
                for (int gi = 0; gi < 2; ++gi) {
                    const int blocks = gi == 0 ? 1280 : 1024, ipw = (199680 + blocks * 4 - 1) / (blocks * 4);
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipEventRecord(a));
                        for (int k = 0; k < 10; ++k) {
#define L2(M, D) mix2<M, D><<<blocks, 256>>>(rec, nrec, dw, col, nseg, ipw, out, order)
                            if (dense == 0) { if (mode == 0) L2(0, 0); if (mode == 1) L2(1, 0); if (mode == 2) L2(2, 0); }
                            else { if (mode == 0) L2(0, 1); if (mode == 1) L2(1, 1); if (mode == 2) L2(2, 1); }
                        }
                        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                        if (rep > 0 && ms < best) best = ms;
                    }
                    printf("{\"probe\": \"mix2\", \"address_order\": \"%s\", \"volume_access\": \"%s\", \"mode\": \"%s\", \"workgroups\": %d, \"us_per_launch\": %.1f}\n",
                           order ? "kernel" : "random", dense ? "dense_batches_2x32" : "items_34_of_64", names[mode], blocks, best * 100.0);
                }
    // round 4: 16 rows x 4 k per wavefront (tile pattern)
    for (int drift = 0; drift < 3; ++drift)
        for (int mode = 0; mode < 3; ++mode)
            for (int gi = 0; gi < 2; ++gi) {
                const int blocks = gi == 0 ? 1280 : 1024, steps = (199680 + blocks * 4 - 1) / (blocks * 4);
                const int drift16 = drift == 0 ? 16 * 16 / 8 : (drift == 1 ? 16 * 16 * 5 / 16 : 16 * 16);      // 2, 5 or 16 pixel columns per 16 rows
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(a));
                    for (int k = 0; k < 10; ++k) {
                        if (mode == 0) mix3<0><<<blocks, 256>>>(rec, nrec, dw, col, 512u * 512u, steps, out, drift16);
                        if (mode == 1) mix3<1><<<blocks, 256>>>(rec, nrec, dw, col, 512u * 512u, steps, out, drift16);
                        if (mode == 2) mix3<2><<<blocks, 256>>>(rec, nrec, dw, col, 512u * 512u, steps, out, drift16);
                    }
                    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("{\"probe\": \"mix3\", \"pattern\": \"16_rows_x_4_k_per_wavefront\", \"pixel_columns_per_16_rows\": %d, \"mode\": \"%s\", \"workgroups\": %d, \"us_per_launch\": %.1f}\n",
                       drift16 / 16, names[mode], blocks, best * 100.0);
            }
    return 0;
}
