// tests/emu/emu_engine.cpp -- the launch engine of the host emulation (tests/emu/hip/hip_runtime.h): workgroups one
// after the other, one host thread per lane, pthread barriers for the wave / workgroup rendezvous.
#include "hip/hip_runtime.h"

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;

namespace {
struct Wave {
    pthread_barrier_t bar;
    alignas(64) unsigned x[64 * 32];
};
struct Block {
    pthread_barrier_t bar;
    std::vector<Wave> waves;
};
std::mutex g_mu;
std::map<std::string, int> g_counts;  // launches by kernel expression, e.g. "(k_cqt_octave_f16<H, R12, TIMING>)"
thread_local Wave *t_wave;
thread_local Block *t_block;
thread_local int t_lane;
}  // namespace

namespace emu {
void wave_barrier() { pthread_barrier_wait(&t_wave->bar); }
void block_barrier() { pthread_barrier_wait(&t_block->bar); }
unsigned *exchange() { return t_wave->x; }
int lane() { return t_lane; }

void launch(const char *kernel, dim3 grid, dim3 block, const std::function<void()> &body) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_counts[kernel];
    }
    const unsigned n = block.x * block.y * block.z;
    if (n == 0 || n % 64 || block.y != 1 || block.z != 1) {
        fprintf(stderr, "emu: block of %u x %u x %u threads not modelled\n", block.x, block.y, block.z);
        abort();
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                Block blk;
                blk.waves = std::vector<Wave>(n / 64);
                pthread_barrier_init(&blk.bar, nullptr, n);
                for (Wave &w : blk.waves) pthread_barrier_init(&w.bar, nullptr, 64);
                std::vector<std::thread> lanes;
                lanes.reserve(n);
                for (unsigned t = 0; t < n; ++t)
                    lanes.emplace_back([&, t] {
                        threadIdx = EmuIdx{t, 0, 0};
                        blockIdx = EmuIdx{bx, by, bz};
                        blockDim = EmuIdx{block.x, block.y, block.z};
                        gridDim = EmuIdx{grid.x, grid.y, grid.z};
                        t_block = &blk;
                        t_wave = &blk.waves[t / 64];
                        t_lane = (int)(t % 64);
                        body();
                    });
                for (std::thread &th : lanes) th.join();
                for (Wave &w : blk.waves) pthread_barrier_destroy(&w.bar);
                pthread_barrier_destroy(&blk.bar);
            }
}
}  // namespace emu

// launches so far of kernels whose launch expression contains `part` (read by the tests through ctypes)
extern "C" int afx_emulated_launches(const char *part) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (const auto &kv : g_counts)
        if (kv.first.find(part) != std::string::npos) n += kv.second;
    return n;
}
