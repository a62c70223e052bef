// Guard-page device allocator for the GPU tests (test infrastructure, not part of the product library).
//
// Plugged into torch with torch.cuda.memory.CUDAPluggableAllocator (tests/guard/__init__.py) when NERFAIL_GUARD_ALLOC=1:
// every tensor becomes its own HIP virtual-memory mapping whose END coincides (to NERFAIL_GUARD_ALIGN bytes, default 16)
// with the end of the mapped range, and the address range behind it is reserved but never mapped. A kernel that reads or
// writes one element past the end of any tensor therefore faults deterministically ("Memory access fault by GPU") instead
// of depending on what the caching allocator happens to have mapped there. A freed tensor is unmapped at once (after a
// device synchronise), so a launch that was handed the pointer of a temporary faults as well.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/types.h>
#include <mutex>
#include <unordered_map>

namespace {

struct Block { hipDeviceptr_t va; size_t mapped; hipMemGenericAllocationHandle_t h; };
// Address space: every tensor has its own reservation (tensor + one granule), which is NEVER freed - a freed tensor's
// addresses stay unmapped for the rest of the process (use-after-free faults for good), and the HIP runtime never sees an
// address range freed and reserved again (with hipMemAddressFree per tensor, hipMemcpy H2D into a later tensor at a recycled
// address silently dropped data on ROCm 7.2: tools/debug/guard_index_repro.py; mapping sub-ranges of one big reservation
// is refused by hipMemSetAccess).
std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
size_t g_gran = 0, g_align = 16, g_live = 0, g_peak = 0, g_count = 0;

void die(const char* what, hipError_t e) {
    fprintf(stderr, "[guard_alloc] %s failed: %s\n", what, hipGetErrorString(e));
    abort();
}
#define CK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) die(#call, e__); } while (0)

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" void* nf_guard_malloc(ssize_t size, int device, hipStream_t) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (g_gran == 0) {
        CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityRecommended));
        const char* a = getenv("NERFAIL_GUARD_ALIGN");
        if (a && atol(a) > 0) g_align = (size_t)atol(a);
        fprintf(stderr, "[guard_alloc] active: granularity %zu B, tensor ends aligned to %zu B\n", g_gran, g_align);
    }
    Block b;
    b.mapped = round_up((size_t)size, g_gran);
    const size_t span = b.mapped + g_gran;                // one granule of unmapped addresses behind every tensor
    CK(hipMemAddressReserve(&b.va, span, g_gran, nullptr, 0));
    CK(hipMemCreate(&b.h, b.mapped, &prop, 0));
    CK(hipMemMap(b.va, b.mapped, 0, b.h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, b.mapped, &acc, 1));
    void* user = (char*)b.va + b.mapped - round_up((size_t)size, g_align);
    g_blocks[user] = b;
    g_live += b.mapped;
    if (g_live > g_peak) g_peak = g_live;
    ++g_count;
    return user;
}

extern "C" void nf_guard_free(void* ptr, ssize_t, int, hipStream_t) {
    if (!ptr) return;
    CK(hipDeviceSynchronize());                           // work already enqueued may still use the block
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_blocks.find(ptr);
    if (it == g_blocks.end()) {
        fprintf(stderr, "[guard_alloc] free of unknown pointer %p\n", ptr);
        abort();
    }
    Block b = it->second;
    g_blocks.erase(it);
    CK(hipMemUnmap(b.va, b.mapped));
    CK(hipMemRelease(b.h));
    g_live -= b.mapped;
}

extern "C" void nf_guard_stats(size_t* count, size_t* live, size_t* peak) {
    std::lock_guard<std::mutex> lk(g_mu);
    *count = g_count; *live = g_live; *peak = g_peak;
}
