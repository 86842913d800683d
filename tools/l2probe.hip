// What the per-XCD L2s of an MI355X do with write-back lines that two XCDs touch -- the facts the FFM row store policy rests on
// (kernels.hip "store policy").  Standalone: hipcc --offload-arch=gfx950 -O2 tools/l2probe.hip -o tools/l2probe && tools/l2probe
//
// Two single-wave workgroups, A on XCD 0 and B on XCD 1 (s_getreg HW_REG_XCC_ID), take turns through a device-scope turn counter.
// Every scenario works on its own 64 lines of 128 B (lane l of a wave owns the 16-byte chunk (l & 7) of line (l >> 3) + 8 * pass).
//
//   S1  clean copy:   A plain-stores 1, writes its L2 back (buffer_wbl2 sc1), re-reads (line stays valid and clean in A's L2);
//                     B sc1-stores 2;  A sc1-loads         -> does A see 2 (fresh) or its own clean copy 1 (stale)?
//   S1p the same with B storing plain + buffer_wbl2 sc1 instead of sc1 stores.
//   S2  dirty copy:   A plain-stores 1 (dirty, no write-back);  B sc1-loads (sees 0 = memory or 1?), then sc1-stores 2;
//                     A sc1-loads (own dirty 1, or 2?), then writes back;  B sc1-loads -> 1 (A's late write-back wins) or 2?
//   S3  partial line: A plain-stores chunk 0 of every line (the rest of the line untouched);  B sc1-stores chunk 1;  A writes back;
//                     B sc1-loads the lines -> chunk 1 still 2 (the write-back wrote A's dirty BYTES) or 0 (it wrote A's whole, stale LINE)?
//   S3r the same after A has first READ the whole lines (so its L2 holds them valid before B's store).
//   S4  eviction:     A plain-stores 1 to the lines, then streams 32 MB through its L2 with plain loads and stores (no write-back instruction);
//                     B sc1-loads -> has the dirty data reached memory by eviction alone?
//   S5  sc1 load of an own dirty line: A plain-stores 1, then sc1-loads the lines (no write-back);  B sc1-loads -> did A's load push the line out?
//   S6  sc1 store into an own dirty line: A plain-stores chunk 0 = 1, then sc1-stores chunk 1 = 5 of the same lines (no write-back);
//                     B sc1-loads -> chunk 1 is 5; is chunk 0 out as well (the write-through took the dirty bytes along) or still private to A?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int kAuxPlain = 0, kAuxSc1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ void st(unsigned *base, unsigned bytes, unsigned off, unsigned v) {
    u4 x = {v, v, v, v};
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc(base, bytes), (int)off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ unsigned ld(const unsigned *base, unsigned bytes, unsigned off) {
    u4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc(base, bytes), (int)off, 0, AUX);
    return x[0];
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wbl2() {
    asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ void wait_turn(unsigned *turn, unsigned want) {
    while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void pass_turn(unsigned *turn, unsigned next) {
    drain();
    __hip_atomic_store(turn, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned kLines = 64, kLineBytes = 128, kRegion = kLines * kLineBytes;  // 8 KB per scenario
constexpr int kScen = 8;

struct Ctl {
    unsigned turn;       // scenario * 16 + step
    int role_a, role_b;  // claimed block ids
    unsigned res[kScen][8];
};

// counts lanes whose value equals v (one wave)
__device__ __forceinline__ unsigned count_eq(unsigned x, unsigned v) { return (unsigned)__popcll(__ballot(x == v)); }

__global__ void probe(Ctl *c, unsigned *data, unsigned *scratch, size_t scratch_words) {
    const unsigned xcc = xcc_id();
    const int lane = threadIdx.x;
    int role = -1;
    if (lane == 0) {
        if (xcc == 0 && atomicCAS(&c->role_a, -1, (int)blockIdx.x) == -1) role = 0;
        if (xcc == 1 && atomicCAS(&c->role_b, -1, (int)blockIdx.x) == -1) role = 1;
    }
    role = __shfl(role, 0, 64);
    if (role < 0) return;
    const bool A = role == 0;
    // lane's chunk: 8 passes of 8 lines
    auto for_lines = [&](auto fn) {
        for (unsigned p = 0; p < 8; ++p) fn((p * 8 + (lane >> 3)) * kLineBytes + (lane & 7) * 16);
    };
    for (int sc = 0; sc < kScen; ++sc) {
        unsigned *X = data + (size_t)sc * (kRegion / 4);
        unsigned *turn = &c->turn;
        const unsigned t0 = sc * 16;
        unsigned *res = c->res[sc];
        if (sc == 0 || sc == 1) {  // S1 / S1p
            if (A) {
                wait_turn(turn, t0 + 0);
                for_lines([&](unsigned o) { st<kAuxPlain>(X, kRegion, o, 1u); });
                wbl2();
                unsigned own = 0;
                for_lines([&](unsigned o) { own += count_eq(ld<kAuxSc1>(X, kRegion, o), 1u); });
                if (lane == 0) res[0] = own;  // (512 = A reads its own data back)
                pass_turn(turn, t0 + 1);
                wait_turn(turn, t0 + 2);
                unsigned fresh = 0, stale = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    fresh += count_eq(v, 2u);
                    stale += count_eq(v, 1u);
                });
                if (lane == 0) { res[1] = fresh; res[2] = stale; }
                pass_turn(turn, t0 + 16);
            } else {
                wait_turn(turn, t0 + 1);
                if (sc == 0) {
                    for_lines([&](unsigned o) { st<kAuxSc1>(X, kRegion, o, 2u); });
                    drain();
                } else {
                    for_lines([&](unsigned o) { st<kAuxPlain>(X, kRegion, o, 2u); });
                    wbl2();
                }
                pass_turn(turn, t0 + 2);
            }
        } else if (sc == 2) {  // S2
            if (A) {
                wait_turn(turn, t0 + 0);
                for_lines([&](unsigned o) { st<kAuxPlain>(X, kRegion, o, 1u); });
                pass_turn(turn, t0 + 1);
                wait_turn(turn, t0 + 2);
                unsigned own = 0, other = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    own += count_eq(v, 1u);
                    other += count_eq(v, 2u);
                });
                if (lane == 0) { res[2] = own; res[3] = other; }
                wbl2();
                pass_turn(turn, t0 + 3);
            } else {
                wait_turn(turn, t0 + 1);
                unsigned zero = 0, one = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    zero += count_eq(v, 0u);
                    one += count_eq(v, 1u);
                });
                if (lane == 0) { res[0] = zero; res[1] = one; }
                for_lines([&](unsigned o) { st<kAuxSc1>(X, kRegion, o, 2u); });
                pass_turn(turn, t0 + 2);
                wait_turn(turn, t0 + 3);
                unsigned a_wins = 0, b_stays = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    a_wins += count_eq(v, 1u);
                    b_stays += count_eq(v, 2u);
                });
                if (lane == 0) { res[4] = a_wins; res[5] = b_stays; }
                pass_turn(turn, t0 + 16);
            }
        } else if (sc == 3 || sc == 4) {  // S3 / S3r
            if (A) {
                wait_turn(turn, t0 + 0);
                if (sc == 4) {
                    unsigned z = 0;
                    for_lines([&](unsigned o) { z += ld<kAuxPlain>(X, kRegion, o); });
                    if (z == 0x12345678u) res[7] = z;  // (keeps the loads)
                    drain();
                }
                pass_turn(turn, t0 + 1);
                wait_turn(turn, t0 + 2);
                for_lines([&](unsigned o) { if ((lane & 7) == 0) st<kAuxPlain>(X, kRegion, o, 1u); });
                drain();
                wbl2();
                pass_turn(turn, t0 + 3);
            } else {
                wait_turn(turn, t0 + 1);
                for_lines([&](unsigned o) { if ((lane & 7) == 1) st<kAuxSc1>(X, kRegion, o, 2u); });
                pass_turn(turn, t0 + 2);
                wait_turn(turn, t0 + 3);
                unsigned c0 = 0, c1 = 0, c1lost = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    c0 += (unsigned)__popcll(__ballot((lane & 7) == 0 && v == 1u));
                    c1 += (unsigned)__popcll(__ballot((lane & 7) == 1 && v == 2u));
                    c1lost += (unsigned)__popcll(__ballot((lane & 7) == 1 && v == 0u));
                });
                if (lane == 0) { res[0] = c0; res[1] = c1; res[2] = c1lost; }  // 64 lines each
                pass_turn(turn, t0 + 16);
            }
        } else if (sc == 6 || sc == 7) {  // S5 / S6
            if (A) {
                wait_turn(turn, t0 + 0);
                for_lines([&](unsigned o) { if (sc == 6 || (lane & 7) == 0) st<kAuxPlain>(X, kRegion, o, 1u); });
                drain();
                if (sc == 6) {
                    unsigned z = 0;
                    for_lines([&](unsigned o) { z += ld<kAuxSc1>(X, kRegion, o); });
                    if (z == 0x12345678u) res[7] = z;
                } else {
                    for_lines([&](unsigned o) { if ((lane & 7) == 1) st<kAuxSc1>(X, kRegion, o, 5u); });
                }
                drain();
                pass_turn(turn, t0 + 1);
            } else {
                wait_turn(turn, t0 + 1);
                unsigned c0 = 0, c1 = 0, all1 = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    all1 += count_eq(v, 1u);
                    c0 += (unsigned)__popcll(__ballot((lane & 7) == 0 && v == 1u));
                    c1 += (unsigned)__popcll(__ballot((lane & 7) == 1 && v == 5u));
                });
                if (lane == 0) { res[0] = all1; res[1] = c0; res[2] = c1; }
                pass_turn(turn, t0 + 16);
            }
        } else {  // S4
            if (A) {
                wait_turn(turn, t0 + 0);
                for_lines([&](unsigned o) { st<kAuxPlain>(X, kRegion, o, 1u); });
                drain();
                // stream through this XCD's L2 (4 MB): plain loads + plain stores of a scratch buffer, no write-back instruction
                unsigned acc = 0;
                for (size_t i = lane; i < scratch_words; i += 64) {
                    acc += scratch[i];
                    scratch[i] = acc;
                }
                if (acc == 0x12345678u) res[7] = acc;
                drain();
                pass_turn(turn, t0 + 1);
            } else {
                wait_turn(turn, t0 + 1);
                unsigned seen = 0, not_yet = 0;
                for_lines([&](unsigned o) {
                    const unsigned v = ld<kAuxSc1>(X, kRegion, o);
                    seen += count_eq(v, 1u);
                    not_yet += count_eq(v, 0u);
                });
                if (lane == 0) { res[0] = seen; res[1] = not_yet; }
                pass_turn(turn, t0 + 16);
            }
        }
    }
}

int main() {
    Ctl *c;
    unsigned *data, *scratch;
    const size_t scratch_words = (32u << 20) / 4;
    hipMalloc(&c, sizeof(Ctl));
    hipMalloc(&data, (size_t)kScen * kRegion);
    hipMalloc(&scratch, scratch_words * 4);
    int ok_runs = 0;
    for (int rep = 0; rep < 3; ++rep) {
        Ctl h{};
        h.role_a = h.role_b = -1;
        hipMemcpy(c, &h, sizeof(h), hipMemcpyHostToDevice);
        hipMemset(data, 0, (size_t)kScen * kRegion);
        hipMemset(scratch, 0, scratch_words * 4);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(64), dim3(64), 0, 0, c, data, scratch, scratch_words);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("launch failed\n");
            return 1;
        }
        hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
        if (h.role_a < 0 || h.role_b < 0) {
            printf("rep %d: no workgroup on XCD 0 / 1 (a=%d b=%d)\n", rep, h.role_a, h.role_b);
            continue;
        }
        ok_runs++;
        printf("rep %d (A = block %d on XCD 0, B = block %d on XCD 1)\n", rep, h.role_a, h.role_b);
        printf("  S1  clean copy, B stores sc1        : A re-reads own %u/512; after B's store A sees fresh %u, stale %u of 512\n", h.res[0][0], h.res[0][1], h.res[0][2]);
        printf("  S1p clean copy, B stores plain+wbl2 : A re-reads own %u/512; after B's store A sees fresh %u, stale %u of 512\n", h.res[1][0], h.res[1][1], h.res[1][2]);
        printf("  S2  dirty copy: B before its store sees memory(0) %u, A's dirty(1) %u; A then sees own(1) %u, B's(2) %u; after A's write-back B sees A's %u, its own %u (of 512)\n",
               h.res[2][0], h.res[2][1], h.res[2][2], h.res[2][3], h.res[2][4], h.res[2][5]);
        printf("  S3  partial line (A never read it)  : A's chunk arrived %u/64, B's chunk kept %u, B's chunk zeroed %u\n", h.res[3][0], h.res[3][1], h.res[3][2]);
        printf("  S3r partial line (A had read it)    : A's chunk arrived %u/64, B's chunk kept %u, B's chunk zeroed %u\n", h.res[4][0], h.res[4][1], h.res[4][2]);
        printf("  S4  eviction only (32 MB streamed)  : B sees A's data %u, still memory's %u (of 512)\n", h.res[5][0], h.res[5][1]);
        printf("  S5  A sc1-loads its own dirty lines : B then sees A's data in %u of 512 chunks\n", h.res[6][0]);
        printf("  S6  A sc1-stores into its dirty line: B sees the sc1-stored chunk in %u/64 lines, the plain-stored (dirty) chunk in %u/64\n", h.res[7][2], h.res[7][1]);
    }
    return ok_runs ? 0 : 2;
}
