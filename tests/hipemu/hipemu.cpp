// hipemu runtime: fibers + block/wave rendezvous (TEST INFRASTRUCTURE, see hip_runtime.h).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hipemu {

thread_local ThreadCtx tls;
int g_dma_late = 0;

namespace {

constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

enum State : int { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct FiberImpl : Fiber {
    void* sp = nullptr;
    State st = DONE;
    char* stack = nullptr;
};

struct WaveBuf {
    alignas(64) char slot[2][64][64];
    int parity = 0;
    int waiting = 0;
};

struct Worker {
    std::vector<FiberImpl> fibers;
    std::vector<WaveBuf> waves;
    void* sched_sp = nullptr;
    const std::function<void()>* body = nullptr;
    std::vector<char> dyn;
    void* dyn_exact = nullptr;
    int nthreads = 0, nwaves = 0;
    Worker() : fibers(kMaxThreads), waves(kMaxThreads / 64) {
        for (auto& f : fibers) {
            f.stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                                  MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (f.stack == MAP_FAILED) { perror("hipemu mmap"); abort(); }
        }
    }
};

thread_local Worker* worker = nullptr;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

void yield_to_sched() {
    FiberImpl* f = (FiberImpl*)tls.cur;
    hipemu_switch(&f->sp, worker->sched_sp);
}

void fiber_entry() {
    FiberImpl* f = (FiberImpl*)tls.cur;
    (*worker->body)();
    f->st = DONE;
    yield_to_sched();
    abort();  // never resumed
}

void prepare(FiberImpl& f) {
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** s = (void**)(top - 64);
    for (int i = 0; i < 6; ++i) s[i] = nullptr;
    s[6] = (void*)&fiber_entry;  // return address, at top-16
    s[7] = nullptr;
    f.sp = (void*)s;
    f.st = RUN;
    f.npend = 0;
}

void run_block(Worker& w, dim3 block) {
    const int nt = w.nthreads;
    int t = 0;
    for (unsigned z = 0; z < block.z; ++z)
        for (unsigned y = 0; y < block.y; ++y)
            for (unsigned x = 0; x < block.x; ++x, ++t) {
                FiberImpl& f = w.fibers[t];
                f.tid = {x, y, z};
                f.linear = t; f.wave = t / 64; f.lane = t % 64;
                prepare(f);
            }
    for (int i = 0; i < w.nwaves; ++i) { w.waves[i].parity = 0; w.waves[i].waiting = 0; }
    int live = nt;
    while (live > 0) {
        bool progress = false;
        int at_block = 0;
        for (int wv = 0; wv < w.nwaves; ++wv) {
            const int lo = wv * 64, hi = lo + 64 < nt ? lo + 64 : nt;
            for (;;) {  // keep running this wave while wave-level rendezvous complete
                bool ran = false;
                for (int i = lo; i < hi; ++i) {
                    FiberImpl& f = w.fibers[i];
                    if (f.st != RUN) continue;
                    tls.cur = &f;
                    hipemu_switch(&w.sched_sp, f.sp);
                    ran = progress = true;
                    if (f.st == DONE) --live;
                }
                int nlive = 0, nwave = 0;
                for (int i = lo; i < hi; ++i) {
                    State s = w.fibers[i].st;
                    nlive += s != DONE; nwave += s == WAIT_WAVE;
                }
                if (nlive > 0 && nwave == nlive) {  // release the wave
                    for (int i = lo; i < hi; ++i)
                        if (w.fibers[i].st == WAIT_WAVE) w.fibers[i].st = RUN;
                    w.waves[wv].parity ^= 1;
                    progress = true;
                    continue;
                }
                if (!ran) break;
            }
        }
        for (int i = 0; i < nt; ++i) at_block += w.fibers[i].st == WAIT_BLOCK;
        if (live > 0 && at_block == live) {
            for (int i = 0; i < nt; ++i)
                if (w.fibers[i].st == WAIT_BLOCK) w.fibers[i].st = RUN;
            progress = true;
        }
        if (!progress && live > 0) {
            int nw = 0, nb = 0;
            for (int i = 0; i < nt; ++i) { nw += w.fibers[i].st == WAIT_WAVE; nb += w.fibers[i].st == WAIT_BLOCK; }
            fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d live, %d at __syncthreads, %d at wave op "
                    "(divergent barrier or wave op under divergence)\n",
                    tls.bid.x, tls.bid.y, tls.bid.z, live, nb, nw);
            abort();
        }
    }
}

// ---- persistent worker pool -------------------------------------------------------------
struct Job {
    dim3 grid, block;
    size_t shmem;
    const std::function<void()>* body;
    std::atomic<long> next{0};
    long total;
    std::atomic<int> remaining{0};
};

struct Pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    Job* job = nullptr;
    unsigned long gen = 0;
    int nworkers;
    Pool() {
        const char* e = getenv("HIPEMU_THREADS");
        nworkers = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (nworkers < 1) nworkers = 1;
        for (int i = 0; i < nworkers; ++i) threads.emplace_back([this] { loop(); });
        for (auto& t : threads) t.detach();
    }
    void work(Job* j) {
        if (!worker) worker = new Worker();
        Worker& w = *worker;
        w.body = j->body;
        w.nthreads = j->block.x * j->block.y * j->block.z;
        w.nwaves = (w.nthreads + 63) / 64;
        static const bool exact_lds = getenv("HIPEMU_EXACT_LDS") != nullptr;
        if (exact_lds) {
            // sanitizer runs (tools/build_emu_asan.sh): the dynamic LDS of every launch is a heap block of EXACTLY the requested
            // size, so an access past the launch's LDS request is a heap overflow the sanitizer reports (the pooled buffer
            // below keeps whatever the largest earlier launch asked for)
            free(w.dyn_exact);
            w.dyn_exact = nullptr;
            if (j->shmem && posix_memalign(&w.dyn_exact, 64, j->shmem) != 0) abort();
            tls.dyn = (char*)w.dyn_exact;
        } else {
            if (w.dyn.size() < j->shmem + 64) w.dyn.resize(j->shmem + 64);
            tls.dyn = (char*)(((uintptr_t)w.dyn.data() + 63) & ~(uintptr_t)63);
        }
        tls.bdim = {j->block.x, j->block.y, j->block.z};
        tls.gdim = {j->grid.x, j->grid.y, j->grid.z};
        for (;;) {
            long b = j->next.fetch_add(1);
            if (b >= j->total) break;
            tls.bid.x = (unsigned)(b % j->grid.x);
            tls.bid.y = (unsigned)((b / j->grid.x) % j->grid.y);
            tls.bid.z = (unsigned)(b / ((long)j->grid.x * j->grid.y));
            run_block(w, j->block);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return gen != seen; });
                seen = gen; j = job;
            }
            work(j);
            if (j->remaining.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(mu);
                done_cv.notify_all();
            }
        }
    }
    void run(Job& j) {
        j.remaining = nworkers;
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &j; ++gen;
        }
        cv.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return j.remaining.load() == 0; });
    }
};

Pool& pool() { static Pool* p = new Pool(); return *p; }
std::mutex launch_mu;

}  // namespace

const uint3_& cur_tid() { return tls.cur->tid; }

void block_barrier() {
    ((FiberImpl*)tls.cur)->st = WAIT_BLOCK;
    yield_to_sched();
}

void* wave_exchange(const void* mine, size_t bytes) {
    FiberImpl* f = (FiberImpl*)tls.cur;
    WaveBuf& wb = worker->waves[f->wave];
    int p = wb.parity;
    if (bytes > 64) abort();
    memcpy(wb.slot[p][f->lane], mine, bytes);
    f->st = WAIT_WAVE;
    yield_to_sched();
    return wb.slot[p];
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    std::lock_guard<std::mutex> lk(launch_mu);
    if ((long)block.x * block.y * block.z > kMaxThreads) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    Job j;
    j.grid = grid; j.block = block; j.shmem = shmem; j.body = &body;
    j.total = (long)grid.x * grid.y * grid.z;
    if (j.total == 0) return;
    pool().run(j);
}

}  // namespace hipemu

// tests: 1 = LDS-DMA lands at the issuing lane's covering s_waitcnt (latest), 0 = at issue (earliest)
extern "C" void hipemu_set_dma_late(int on) { hipemu::g_dma_late = on; }
extern "C" int hipemu_get_dma_late() { return hipemu::g_dma_late; }
