// Memory source of a feature database: hands out large, 2 MiB-aligned blocks (advised as huge pages) and keeps them
// until it is destroyed.  The database puts a std::pmr pool on top of it, so the steady stream of small growing
// observation arrays costs neither a system call nor a page fault per few kilobytes, and freed blocks are reused
// inside the database instead of going back and forth between the process and the kernel.
#pragma once
#include <sys/mman.h>

#include <cstdlib>
#include <memory_resource>
#include <new>
#include <vector>

namespace feature_tracker {

class ChunkResource : public std::pmr::memory_resource {
public:
    // 8 MiB per block: a stream of 150 features fills 2 MiB in a few hundred frames, and streams that start together
    // then all take their next block (a fresh mapping and its first-touch faults) in the same few frame sets, which
    // shows as a millisecond stall of the database writers; larger blocks make that four times rarer
    explicit ChunkResource(size_t chunk_bytes = size_t(8) << 20) : chunk(chunk_bytes) {}
    ChunkResource(const ChunkResource&) = delete;
    ChunkResource& operator=(const ChunkResource&) = delete;
    ~ChunkResource() override {
        for (void* p : blocks) std::free(p);
    }
    size_t bytes_reserved() const { return reserved; }

protected:
    void* do_allocate(size_t bytes, size_t alignment) override {
        // requests larger than the bump block get a block of their own
        if (alignment > HUGE) throw std::bad_alloc();
        size_t at = (used + alignment - 1) & ~(alignment - 1);
        if (!cur || at + bytes > cur_size) {
            const size_t want = bytes > chunk ? ((bytes + HUGE - 1) & ~(HUGE - 1)) : chunk;
            void* p = std::aligned_alloc(HUGE, want);
            if (!p) throw std::bad_alloc();
            (void)madvise(p, want, MADV_HUGEPAGE);   // advisory: ignore failure
            blocks.push_back(p);
            reserved += want;
            cur = static_cast<char*>(p); cur_size = want; at = 0;
        }
        used = at + bytes;
        return cur + at;
    }
    void do_deallocate(void*, size_t, size_t) override {}   // blocks live until destruction; the pool above reuses memory
    bool do_is_equal(const std::pmr::memory_resource& o) const noexcept override { return this == &o; }

private:
    static constexpr size_t HUGE = size_t(2) << 20;
    size_t chunk;
    std::vector<void*> blocks;
    char* cur = nullptr;
    size_t cur_size = 0, used = 0, reserved = 0;
};

}  // namespace feature_tracker
