// K-loop structure experiment (kernel-development aid, not part of lib2dod_hip.so): C[M][N] = A[M][K] * B[N][K]^T in bf16 with
// fp32 accumulation, 128 x 64 output tiles, 64-deep K slices staged by LDS-DMA into an XOR-swizzled ring -- the inner loop of
// conv_tile.hip for a 1x1 convolution -- in two forms:
//   variant 0 "sync":  all 8 waves issue their share of a slice's DMA, then compute; counted vmcnt + one s_barrier per slice
//                      (what conv_tile_kernel does)
//   variant 1 "roles": waves 0-3 only load (6 DMA instructions each per slice, never behind a barrier), waves 4-7 only compute
//                      (64 x 32 each); per ring slot a FULL and a FREE arrival counter in LDS, no s_barrier in the loop
// Question: does the CU's LDS-DMA path take in more than the ~37 GB/s it gets in the sync form when the loaders never stall?
// Every spin is bounded: a protocol error ends the kernel with err[0] != 0 instead of hanging the GPU.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libkloop_bench.so tools/kloop_bench.hip     run: tools/kloop_bench.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {

constexpr int BM = 128, BN = 64, BK = 64, T = 512, NW = 8;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, SLICE = A_BYTES + B_BYTES;
constexpr unsigned kOob = 0xFFFFFFF0u;

__device__ __forceinline__ int swz(int chunk, int row) { return chunk ^ (row & 7); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void lds_add_u32(unsigned addr, unsigned v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_read_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

struct Params {
    const bf16_t* A; const bf16_t* B; float* C; unsigned* err;
    int M, N, K, tiles_n;
    unsigned a_bytes, b_bytes;
};

__device__ __forceinline__ void tile_of(const Params& p, int& m0, int& n0) {
    // XCD-chunked unit list, n fastest (as conv_tile.hip)
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    m0 = (unit / p.tiles_n) * BM;
    n0 = (unit % p.tiles_n) * BN;
}

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// ------------------------------------------------------------------ variant 0: barrier-synchronous, 8 waves load + compute
template <int S>
__global__ __launch_bounds__(512, 2) void kloop_sync(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    constexpr int WM = 4, WN = 2, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16, KK = BK / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int frow = lane & 15, fchunk = lane >> 4;
    int m0, n0;
    tile_of(p, m0, n0);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned a_voff[2], b_voff[1];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + NW * i) * 8 + lrow;
        a_voff[i] = m0 + r < p.M ? (unsigned)(m0 + r) * (unsigned)(p.K * 2) + (unsigned)swz(lslot, r) * 16u : kOob;
    }
    {
        const int r = wave * 8 + lrow;
        b_voff[0] = n0 + r < p.N ? (unsigned)(n0 + r) * (unsigned)(p.K * 2) + (unsigned)swz(lslot, r) * 16u : kOob;
    }
    unsigned soff = 0;
    auto issue = [&](const int slot) {
        unsigned char* sa = ring + slot * A_BYTES;
        unsigned char* sb = ring + S * A_BYTES + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sa + (wave + NW * i) * 1024), 16, a_voff[i], soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb + wave * 1024), 16, b_voff[0], soff, 0, 0);
        soff += BK * 2;
    };
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned a_foff[KK], b_foff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a_foff[kk] = (unsigned)((wm * WTM + frow) * (BK * 2) + swz(kk * 4 + fchunk, frow) * 16);
        b_foff[kk] = (unsigned)((wn * WTN + frow) * (BK * 2) + swz(kk * 4 + fchunk, frow) * 16);
    }
    auto mfma_slice = [&](const int slot) {
        const unsigned char* cA = ring + slot * A_BYTES;
        const unsigned char* cB = ring + S * A_BYTES + slot * B_BYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[kk]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = p.K / BK;
    for (int s = 0; s < S - 1 && s < nk; ++s) issue(s);
    int slot = 0;
    for (int k = 0; k < nk; ++k) {
        if (k + S - 1 <= nk) WAIT_VM((S - 2) * 3);
        else WAIT_VM(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (k + S - 1 < nk) issue(slot == 0 ? S - 1 : slot - 1);
        mfma_slice(slot);
        slot = slot + 1 == S ? 0 : slot + 1;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int m = m0 + wm * WTM + i * 16 + frow, n = n0 + wn * WTN + j * 16 + fchunk * 4;
            if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.N + n) = acc[i][j];
        }
}

// ------------------------------------------------------------------ variant 1: loader waves / consumer waves
// LDS: [S A tiles][S B tiles][full[S]][free[S]] -- full[s] counts loader arrivals (4 per fill), free[s] consumer releases (4 per use)
template <int S, int D>
__global__ __launch_bounds__(512, 2) void kloop_roles(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    constexpr int NL = 4, NC = 4;                    // loader / consumer waves
    constexpr int WM = 2, WN = 2, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16, KK = BK / 32;
    constexpr int A_PER = (A_BYTES / 1024) / NL, B_PER = (B_BYTES / 1024) / NL, PER = A_PER + B_PER;      // 4 + 2 DMA instructions per loader and slice
    static_assert(D >= 1 && D < S, "D slices in flight per loader, one slot being consumed");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* flags = reinterpret_cast<unsigned*>(ring + S * SLICE);
    const unsigned full_a = lds_addr(flags), free_a = lds_addr(flags + S);
    if (tid < 2 * S) flags[tid] = 0u;
    __syncthreads();
    int m0, n0;
    tile_of(p, m0, n0);
    const int nk = p.K / BK;
    constexpr int SPIN_MAX = 1 << 22;
    if (wave < NL) {
        // ---------------- loader
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
        const int lrow = lane >> 3, lslot = lane & 7;
        unsigned a_voff[A_PER], b_voff[B_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int r = (wave + NL * i) * 8 + lrow;
            a_voff[i] = m0 + r < p.M ? (unsigned)(m0 + r) * (unsigned)(p.K * 2) + (unsigned)swz(lslot, r) * 16u : kOob;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int r = (wave + NL * i) * 8 + lrow;
            b_voff[i] = n0 + r < p.N ? (unsigned)(n0 + r) * (unsigned)(p.K * 2) + (unsigned)swz(lslot, r) * 16u : kOob;
        }
        unsigned soff = 0;
        int slot = 0, gen = 0;                       // slice k goes to slot k % S as that slot's fill number gen = k / S
        int pslot = 0;                               // slot of the oldest unpublished slice
        for (int k = 0; k < nk; ++k) {
            if (gen > 0) {                           // the slot's previous contents must have been read by all consumers
                const unsigned want = (unsigned)(NC * gen);
                int spin = 0;
                while (lds_read_u32(free_a + 4 * slot) < want) {
                    if (++spin > SPIN_MAX) { if (lane == 0) atomicAdd(p.err, 1u); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            unsigned char* sa = ring + slot * A_BYTES;
            unsigned char* sb = ring + S * A_BYTES + slot * B_BYTES;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sa + (wave + NL * i) * 1024), 16, a_voff[i], soff, 0, 0);
#pragma unroll
            for (int i = 0; i < B_PER; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb + (wave + NL * i) * 1024), 16, b_voff[i], soff, 0, 0);
            soff += BK * 2;
            if (k >= D) {                            // slice k - D has landed once only D slices' pieces are outstanding
                WAIT_VM(D * PER);
                if (lane == 0) lds_add_u32(full_a + 4 * pslot, 1u);
                pslot = pslot + 1 == S ? 0 : pslot + 1;
            }
            if (++slot == S) { slot = 0; ++gen; }
        }
        WAIT_VM(0);
        const int left = nk < D ? nk : D;
        for (int j = 0; j < left; ++j) {
            if (lane == 0) lds_add_u32(full_a + 4 * pslot, 1u);
            pslot = pslot + 1 == S ? 0 : pslot + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    // -------------------- consumer
    const int cw = wave - NL;
    const int wm = cw / WN, wn = cw - wm * WN;
    const int frow = lane & 15, fchunk = lane >> 4;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned a_foff[KK], b_foff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a_foff[kk] = (unsigned)((wm * WTM + frow) * (BK * 2) + swz(kk * 4 + fchunk, frow) * 16);
        b_foff[kk] = (unsigned)((wn * WTN + frow) * (BK * 2) + swz(kk * 4 + fchunk, frow) * 16);
    }
    int slot = 0, gen = 0;
    for (int k = 0; k < nk; ++k) {
        const unsigned want = (unsigned)(NL * (gen + 1));
        int spin = 0;
        while (lds_read_u32(full_a + 4 * slot) < want) {
            if (++spin > SPIN_MAX) { if (lane == 0) atomicAdd(p.err, 1u); return; }
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned char* cA = ring + slot * A_BYTES;
        const unsigned char* cB = ring + S * A_BYTES + slot * B_BYTES;
        bf16x8 af[KK][MI], bfr[KK][NI];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[kk]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slice is in registers: release the slot before computing
        if (lane == 0) lds_add_u32(free_a + 4 * slot, 1u);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        if (++slot == S) { slot = 0; ++gen; }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int m = m0 + wm * WTM + i * 16 + frow, n = n0 + wn * WTN + j * 16 + fchunk * 4;
            if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.N + n) = acc[i][j];
        }
}

template <typename K>
int launch(K kernel, const Params& p, int grid, size_t smem, hipStream_t s) {
    if (smem > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -3;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// variant: 0 sync S=3 | 1 roles S=3 D=1 | 2 roles S=4 D=2 | 3 roles S=5 D=3 | 4 sync S=2
extern "C" int kloop_bench(int variant, const void* A, const void* B, float* C, unsigned* err, int M, int N, int K, void* stream) {
    if (K % BK || N % 8 || (size_t)M * K * 2 >= 0xFFFF0000ull || (size_t)N * K * 2 >= 0xFFFF0000ull) return -1;
    Params p;
    p.A = reinterpret_cast<const bf16_t*>(A); p.B = reinterpret_cast<const bf16_t*>(B); p.C = C; p.err = err;
    p.M = M; p.N = N; p.K = K; p.tiles_n = (N + BN - 1) / BN;
    p.a_bytes = (unsigned)((size_t)M * K * 2); p.b_bytes = (unsigned)((size_t)N * K * 2);
    const int grid = ((M + BM - 1) / BM) * p.tiles_n;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (variant) {
        case 0: return launch(kloop_sync<3>, p, grid, 3 * SLICE, s);
        case 4: return launch(kloop_sync<2>, p, grid, 2 * SLICE, s);
        case 1: return launch(kloop_roles<3, 1>, p, grid, 3 * SLICE + 64, s);
        case 2: return launch(kloop_roles<4, 2>, p, grid, 4 * SLICE + 64, s);
        case 3: return launch(kloop_roles<5, 3>, p, grid, 5 * SLICE + 64, s);
    }
    return -1;
}
