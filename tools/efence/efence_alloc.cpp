// "Electric fence" device allocator for a bounds-checked pass over the GPU test suite (SURVEY 5 asks for a
// sanitizer-equivalent pass, and this image has no ASan-instrumented HIP runtime: /opt/rocm/lib/asan is absent, so -fsanitize=address device
// code has no shadow memory to run against).
//
// Every torch allocation becomes its OWN virtual-memory mapping (hipMemAddressReserve / hipMemCreate / hipMemMap) with an unmapped guard
// granule on both sides, placed so that the tensor ENDS at the end of its last mapped page (start rounded down to 16 bytes): a kernel that
// reads or writes past the end of a tensor -- or before its start by more than the page slack -- touches an unmapped page, the GPU raises a
// memory access fault and the process aborts inside the offending test instead of silently reading a neighbour's bytes.
// Frees synchronise the device first, unmap and release the pages (no caching: use-after-free faults too) and retire the address range.
//
//   hipcc -O2 -shared -fPIC -o tools/efence/libefence.so tools/efence/efence_alloc.cpp
//   SPI_EFENCE=1 python -m pytest tests -m gpu ...        (tests/conftest.py installs it through torch's pluggable-allocator hook)
//
// Test infrastructure only: nothing in spi_amd/ loads it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <mutex>
#include <unordered_map>

namespace {
struct Block { void* va; size_t va_size; size_t mapped; hipMemGenericAllocationHandle_t h; };
std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
size_t g_gran[64] = {0};
unsigned long long g_allocs = 0, g_bytes = 0, g_live = 0, g_peak = 0;

#define EF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[efence] %s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); abort(); } } while (0)

size_t granularity(int device) {
    if (g_gran[device] == 0) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t g = 0;
        EF_CHECK(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum));
        g_gran[device] = g ? g : 4096;
        fprintf(stderr, "[efence] device %d: allocation granularity %zu bytes (guard granule on both sides of every tensor)\n", device, g_gran[device]);
    }
    return g_gran[device];
}
}  // namespace

extern "C" void* efence_malloc(ssize_t size, int device, hipStream_t /*stream*/) {
    if (size <= 0) return nullptr;
    int prev = 0;
    EF_CHECK(hipGetDevice(&prev));
    if (prev != device) EF_CHECK(hipSetDevice(device));
    const size_t g = granularity(device);
    const size_t mapped = ((size_t)size + g - 1) / g * g;
    Block b;
    b.va_size = mapped + 2 * g; b.mapped = mapped;
    EF_CHECK(hipMemAddressReserve(&b.va, b.va_size, g, nullptr, 0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    EF_CHECK(hipMemCreate(&b.h, mapped, &prop, 0));
    char* lo = static_cast<char*>(b.va) + g;
    EF_CHECK(hipMemMap(lo, mapped, 0, b.h, 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    EF_CHECK(hipMemSetAccess(lo, mapped, &acc, 1));
    // the tensor ends where the mapping ends (start rounded DOWN to 16 bytes: kernels use 16-byte vector accesses on tensor bases)
    char* user = lo + ((mapped - (size_t)size) & ~(size_t)15);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_blocks[user] = b;
        ++g_allocs; g_bytes += (size_t)size; g_live += mapped; if (g_live > g_peak) g_peak = g_live;
    }
    if (prev != device) EF_CHECK(hipSetDevice(prev));
    return user;
}

extern "C" void efence_free(void* ptr, ssize_t /*size*/, int device, hipStream_t /*stream*/) {
    if (!ptr) return;
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(ptr);
        if (it == g_blocks.end()) { fprintf(stderr, "[efence] free of an unknown pointer %p\n", ptr); abort(); }
        b = it->second;
        g_blocks.erase(it);
        g_live -= b.mapped;
    }
    int prev = 0;
    EF_CHECK(hipGetDevice(&prev));
    if (prev != device) EF_CHECK(hipSetDevice(device));
    EF_CHECK(hipDeviceSynchronize());                              // kernels queued on any stream may still use the tensor
    char* lo = static_cast<char*>(b.va) + (b.va_size - b.mapped) / 2;
    EF_CHECK(hipMemUnmap(lo, b.mapped));
    EF_CHECK(hipMemRelease(b.h));
    // The address range is NOT returned (hipMemAddressFree): a range that is reserved again right away and mapped to new pages was observed to
    // be read through a stale translation by the next kernel (round 5: the second of two tests saw the first one's camera matrix) -- and a
    // range that is never reused keeps faulting on use-after-free for the rest of the process.  47 bits of address space outlast a test run.
    if (prev != device) EF_CHECK(hipSetDevice(prev));
}

extern "C" void efence_stats(unsigned long long* out4) {
    std::lock_guard<std::mutex> lk(g_mu);
    out4[0] = g_allocs; out4[1] = g_bytes; out4[2] = g_peak; out4[3] = (unsigned long long)g_blocks.size();
}
