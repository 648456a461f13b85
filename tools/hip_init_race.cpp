// Developer tool (GPU box): does host code that allocates and fills multi-megabyte buffers race with the HIP
// runtime's start-up on another thread?  (rt_amd once died 2 times in ~300 runs in a memcpy into a freshly
// allocated texture buffer while a second thread sat in the runtime's initialisation: cli_main.cpp.)
//   hip_init_race overlap|sequential|steady|create [rounds]
// overlap:    thread B makes the process's first HIP calls while thread A allocates, fills, copies and frees buffers
// sequential: the same work after the runtime is up
// steady:     the runtime is up (first calls made on the main thread); then thread B allocates, copies and frees
//             device and pinned memory in a loop while thread A works as above (what a loader thread beside a
//             rendering thread, or rt_create beside it, amounts to)
// create:     as steady, but thread B creates and destroys streams and events and queries occupancy too
// A SIGSEGV prints the faulting address, its si_code (1 = not mapped, 2 = no access) and the lines of
// /proc/self/maps around it, then the exit status is 139.  Build: hipcc -O2 -std=c++17 -pthread (tools/hip_init_race.sh).
#include <hip/hip_runtime.h>

#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>

static void on_segv(int sig, siginfo_t* info, void*) {
    char line[256];
    const unsigned long addr = reinterpret_cast<unsigned long>(info->si_addr);
    int n = snprintf(line, sizeof line, "SIGSEGV at %#lx si_code %d (%s)\n", addr, info->si_code,
                     info->si_code == SEGV_MAPERR ? "address not mapped" : info->si_code == SEGV_ACCERR ? "no access to a mapped page" : "other");
    (void)!write(STDERR_FILENO, line, n);
    // the maps lines whose range ends within 64 MB below or starts within 64 MB above the address
    const int fd = open("/proc/self/maps", O_RDONLY);
    if (fd >= 0) {
        static char buf[1 << 20];
        size_t have = 0;
        ssize_t got;
        while (have < sizeof buf - 1 && (got = read(fd, buf + have, sizeof buf - 1 - have)) > 0) have += static_cast<size_t>(got);
        buf[have] = 0;
        close(fd);
        for (char* p = buf; *p;) {
            char* e = strchr(p, '\n');
            if (!e) break;
            unsigned long lo = 0, hi = 0;
            if (sscanf(p, "%lx-%lx", &lo, &hi) == 2 && hi + (64ul << 20) > addr && lo < addr + (64ul << 20)) (void)!write(STDERR_FILENO, p, e - p + 1);
            p = e + 1;
        }
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char** argv) {
    const bool overlap = argc > 1 && std::strcmp(argv[1], "overlap") == 0;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 40;
    struct sigaction sa {};
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
    std::atomic<bool> up{false};
    auto start_runtime = [&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
            fprintf(stderr, "no device\n");
            _exit(3);
        }
        (void)hipSetDevice(0);
        void* d = nullptr;
        (void)hipMalloc(&d, 1 << 20);
        void* h = nullptr;
        (void)hipHostMalloc(&h, 1 << 20);
        (void)hipDeviceSynchronize();
        up = true;
    };
    std::vector<unsigned char> source(16u << 20);
    for (size_t i = 0; i < source.size(); i++) source[i] = static_cast<unsigned char>(i * 2654435761u >> 24);
    const bool steady = argc > 1 && (std::strcmp(argv[1], "steady") == 0 || std::strcmp(argv[1], "create") == 0);
    const bool create = argc > 1 && std::strcmp(argv[1], "create") == 0;
    std::atomic<bool> stop{false};
    std::thread runtime;
    if (overlap) runtime = std::thread(start_runtime); else start_runtime();
    if (steady) {
        runtime = std::thread([&] {
            std::vector<unsigned char> host(1 << 20, 7);
            while (!stop.load()) {
                void *d = nullptr, *h = nullptr;
                (void)hipMalloc(&d, 8 << 20);
                (void)hipHostMalloc(&h, 2 << 20);
                (void)hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
                (void)hipMemcpyAsync(h, d, host.size(), hipMemcpyDeviceToHost, nullptr);
                if (create) {
                    hipStream_t st;
                    hipEvent_t ev;
                    (void)hipStreamCreate(&st);
                    (void)hipEventCreate(&ev);
                    (void)hipEventRecord(ev, st);
                    (void)hipStreamSynchronize(st);
                    (void)hipEventDestroy(ev);
                    (void)hipStreamDestroy(st);
                }
                (void)hipDeviceSynchronize();
                (void)hipHostFree(h);
                (void)hipFree(d);
            }
        });
    }
    unsigned long sum = 0;
    unsigned seed = static_cast<unsigned>(getpid());
    for (int r = 0; r < rounds || (overlap && !up.load()); r++) {
        seed = seed * 1664525u + 1013904223u;
        const size_t bytes = (4u << 20) + (seed >> 8) % (12u << 20);      // 4..16 MB: an anonymous mmap of its own in glibc's malloc
        std::vector<unsigned char> v(bytes);                               // zero-filled: every page touched
        std::memcpy(v.data(), source.data(), bytes);                       // ... and written again (the loader's decode + copy)
        sum += v[bytes / 2] + v[bytes - 1];
    }
    stop = true;
    if (runtime.joinable()) runtime.join();
    printf("ok %lu\n", sum);
    return 0;
}
