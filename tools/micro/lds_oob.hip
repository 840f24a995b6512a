// Micro-probe (gfx950): what happens to LDS accesses beyond a workgroup's allocation?  The ISA documents them as dropped
// (writes) / zero (reads); K1's scan could then store accepted candidates without clamping the cursor to the list's end.
// Every workgroup fills its own 8 KB of LDS with a pattern, hammers addresses from its allocation's end up to 8 MB with
// stores, waits, and checks (a) its own pattern, (b) what out-of-range reads return.  With several workgroups resident per
// CU a leaking store would corrupt a neighbour.   build: hipcc --offload-arch=gfx950 -O2 lds_oob.hip -o lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kWords = 2048;  // 8 KB per workgroup
typedef __attribute__((address_space(3))) unsigned *lds_u32p;
typedef __attribute__((address_space(3))) unsigned short *lds_u16p;

__global__ __launch_bounds__(256) void probe(unsigned *bad_pattern, unsigned *nonzero_reads, int rounds)
{
    __shared__ unsigned s[kWords];
    const unsigned pat = 0xA5000000u | blockIdx.x;
    for (int k = threadIdx.x; k < kWords; k += 256) s[k] = pat ^ (unsigned)k;
    __syncthreads();
    const unsigned base = (unsigned)(__UINTPTR_TYPE__)(lds_u32p)s;
    unsigned nz = 0;
    for (int r = 0; r < rounds; r++) {
        // from just past the allocation (rounded up to 16 KB so that granule padding is not mistaken for a leak) to 8 MB:
        // far beyond the CU's 160 KB — K1's unclamped cursor can run megabytes past the list, and an address decoder that
        // looked at 16 or 17 bits only would fold those stores back onto the workgroup's own data
        for (unsigned a = base + 16384 + threadIdx.x * 4; a < (8u << 20); a += 1024) {
            *(lds_u32p)(__UINTPTR_TYPE__)a = 0xDEAD0000u | threadIdx.x;
            *(lds_u16p)(__UINTPTR_TYPE__)(a + 2) = (unsigned short)0xBEEF;
            nz += (*(volatile __attribute__((address_space(3))) unsigned *)(__UINTPTR_TYPE__)a != 0u) ? 1u : 0u;
        }
        // ... and the top of the 32-bit range
        *(lds_u32p)(__UINTPTR_TYPE__)(0xFFFFF000u + threadIdx.x * 4) = 0xDEAD0000u | threadIdx.x;
        *(lds_u16p)(__UINTPTR_TYPE__)(0x7FFFF000u + threadIdx.x * 4) = (unsigned short)0xBEEF;
        __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    unsigned bad = 0;
    for (int k = threadIdx.x; k < kWords; k += 256) bad += (s[k] != (pat ^ (unsigned)k)) ? 1u : 0u;
    if (bad) atomicAdd(bad_pattern, bad);
    if (nz) atomicAdd(nonzero_reads, nz);
}

int main()
{
    unsigned *d = nullptr, h[2] = {0, 0};
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    probe<<<256 * 16, 256>>>(d, d + 1, 3);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    std::printf("lds_oob: status %s, corrupted words %u, non-zero out-of-range reads %u  (0 / 0 = stores dropped, reads zero)\n",
                hipGetErrorString(e), h[0], h[1]);
    return (e == hipSuccess && h[0] == 0) ? 0 : 1;
}
