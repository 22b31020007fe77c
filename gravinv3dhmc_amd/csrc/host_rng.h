// libgravhmc host side: the reference's random stream (hmc.py:260,297,95,164).  Included once by gravhmc.hip.
//
// NumPy's legacy RandomState: MT19937, 53-bit doubles from two outputs, the polar Gaussian with its
// cached second value, masked-rejection integers.  Restated here because at C1 / C3 the sampler was
// bound by np.random.randn: 6000 normals per trajectory at ~10-18 ns each on the host against 64 us of
// GPU time.  The stream is NumPy's bit for bit (tests/test_host.py compares draws and states), so it is
// handed back and forth with np.random.get_state / set_state.  What makes it faster than NumPy's
// one-value-at-a-time code while staying sequential where the stream is:
//   * the generator runs RNG_BLOCKS twists ahead and tempers whole blocks (both loops vectorise);
//   * the polar method's rejection is a compaction (every attempt is stored, the slot advances only
//     on acceptance): no data-dependent branch;
//   * sqrt(-2 log(r2) / r2) -- libm's log, as NumPy -- is a second pass over the accepted points,
//     split over GRAVHMC_RNG_THREADS threads (default 4) when the block is large enough.
#pragma once

static const int RNG_BLOCKS = 8;

// The generator's two block loops (624 states per twist, 624 tempered outputs) are most of a draw's sequential time:
// ~10 integer operations per 32-bit output, five outputs per normal.  They vectorise; compiled once more for the wider
// vectors of the host that runs them (the resolver picks at load time; the integer arithmetic is the same bit for bit).
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define GH_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define GH_HOST_CLONES
#endif

struct gh_rng {
    uint32_t keys[RNG_BLOCKS][624];  // generator state after each twist (what np.random.get_state holds)
    uint32_t out[RNG_BLOCKS * 624];  // tempered outputs of those states
    int nb;                          // blocks present
    int p;                           // next output, 0 .. nb * 624
    int has_gauss;
    double gauss;
    std::vector<double> r2;          // squared radii of one call's accepted points
    int threads = 0;                 // threads of a draw (gh_rng_set_threads; 0: GRAVHMC_RNG_THREADS, default 2)
};

GH_HOST_CLONES static void rng_twist(const uint32_t *in, uint32_t *mt)
{
    if (mt != in) memcpy(mt, in, 624 * sizeof(uint32_t));
    int i;
    for (i = 0; i < 624 - 397; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    for (; i < 623; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i - (624 - 397)] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

GH_HOST_CLONES static void rng_temper(const uint32_t *mt, uint32_t *o)
{
    for (int i = 0; i < 624; ++i) {
        uint32_t y = mt[i];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        o[i] = y;
    }
}

// Fewer outputs left than the caller needs (< 624): the last block moves to the front, the rest follow it.
static void rng_refill(gh_rng *r)
{
    const int rem = r->nb * 624 - r->p;
    if (r->nb > 1) {
        memcpy(r->keys[0], r->keys[r->nb - 1], sizeof r->keys[0]);
        memcpy(r->out, r->out + (size_t)(r->nb - 1) * 624, 624 * sizeof(uint32_t));
    }
    for (int j = 1; j < RNG_BLOCKS; ++j) {
        rng_twist(r->keys[j - 1], r->keys[j]);
        rng_temper(r->keys[j], r->out + (size_t)j * 624);
    }
    r->nb = RNG_BLOCKS;
    r->p = 624 - rem;
}

static inline uint32_t rng_next32(gh_rng *r)
{
    if (r->p >= r->nb * 624) rng_refill(r);
    return r->out[r->p++];
}

static inline double rng_from2(uint32_t o0, uint32_t o1)
{
    const int32_t a = (int32_t)(o0 >> 5), b = (int32_t)(o1 >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static inline double rng_double(gh_rng *r)
{
    const uint32_t o0 = rng_next32(r), o1 = rng_next32(r);
    return rng_from2(o0, o1);
}

// `np` accepted points of the polar method: uniform (x1, x2) in the unit disc.  dst gets (x2, x1) per
// point -- the order legacy_gauss returns them in -- and r2 the squared radius.
static void rng_disc_points(gh_rng *r, size_t np, double *dst, double *r2)
{
#pragma clang fp contract(off)
    size_t cnt = 0;
    while (cnt < np) {
        if (r->p + 4 > r->nb * 624) rng_refill(r);
        const uint32_t *o = r->out + r->p;
        r->p += 4;
        const double x1 = 2.0 * rng_from2(o[0], o[1]) - 1.0;
        const double x2 = 2.0 * rng_from2(o[2], o[3]) - 1.0;
        const double q = x1 * x1 + x2 * x2;
        dst[2 * cnt] = x2;
        dst[2 * cnt + 1] = x1;
        r2[cnt] = q;
        cnt += (size_t)((q < 1.0) & (q != 0.0));
    }
}

static inline double rng_disc_factor(double r2)
{
#pragma clang fp contract(off)
    return std::sqrt(-2.0 * std::log(r2) / r2);
}

static void rng_load(gh_rng *r, const uint32_t *key624, int pos, int has_gauss, double cached)
{
    memcpy(r->keys[0], key624, sizeof r->keys[0]);
    rng_temper(r->keys[0], r->out);
    r->nb = 1;
    r->p = pos;
    r->has_gauss = has_gauss ? 1 : 0;
    r->gauss = cached;
}

int gh_rng_create(gh_rng **out, uint32_t seed)
{
    if (!out) return GH_ERR_ARG;
    gh_rng *r = new gh_rng();
    uint32_t key[624];
    key[0] = seed;  // init_genrand: np.random.seed(int)
    for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
    rng_load(r, key, 624, 0, 0.0);
    *out = r;
    return GH_OK;
}

void gh_rng_destroy(gh_rng *r) { delete r; }

// Host cores this process may really use: the affinity mask capped by the cgroup's CPU quota (a container sees all
// cores of its host and is granted a few: a thread team of the visible size is throttled to a crawl).
static int rng_usable_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        long long per = 0;
        if (fscanf(f, "%63s %lld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, atoll(a) / per));
        fclose(f);
    } else if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long quota = -1, per = 0;
        if (fscanf(q, "%lld", &quota) != 1) quota = -1;
        fclose(q);
        if (FILE *pf = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(pf, "%lld", &per) != 1) per = 0;
            fclose(pf);
        }
        if (quota > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, quota / per));
    }
    return std::max(1, n);
}

// threads: helpers of the scale pass (the logarithms: most of a draw's time) for this generator.  0: the default
// (GRAVHMC_RNG_THREADS, else 4 -- right for several generators drawing side by side); -1: as many as the process may
// use, less two for the threads that feed the GPU (ONE chain whose draws are what the GPU waits for).
int gh_host_cores(void) { return rng_usable_cores(); }

int gh_rng_set_threads(gh_rng *r, int threads)
{
    if (!r) return GH_ERR_ARG;
    r->threads = threads < 0 ? std::max(1, std::min(16, rng_usable_cores() - 2)) : threads;
    return GH_OK;
}

int gh_rng_set_state(gh_rng *r, const uint32_t *key624, int pos, int has_gauss, double cached)
{
    if (!r || !key624 || pos < 0 || pos > 624) return GH_ERR_ARG;
    rng_load(r, key624, pos, has_gauss, cached);
    return GH_OK;
}

int gh_rng_get_state(const gh_rng *r, uint32_t *key624, int *pos, int *has_gauss, double *cached)
{
    if (!r || !key624 || !pos || !has_gauss || !cached) return GH_ERR_ARG;
    // NumPy twists lazily: with a block used up it still holds that block's state and pos = 624
    int blk = r->p / 624, at = r->p % 624;
    if (at == 0 && blk > 0) {
        blk -= 1;
        at = 624;
    }
    memcpy(key624, r->keys[blk], sizeof r->keys[0]);
    *pos = at;
    *has_gauss = r->has_gauss;
    *cached = r->gauss;
    return GH_OK;
}

// K trajectories in the reference's order: L = np.random.randint(Lmin, Lmax + 1) (hmc.py:297),
// p0 = np.random.randn(M) * sigma (hmc.py:95), u = np.random.rand() (hmc.py:164).
int gh_rng_draw_trajectories(gh_rng *r, int K, int Lmin, int Lmax, int64_t M, double sigma, int *L, double *p0s,
                             double *us)
{
#pragma clang fp contract(off)
    if (!r || K < 0 || Lmax < Lmin || M < 0 || !L || !p0s || !us) return GH_ERR_ARG;
    const uint32_t rng = (uint32_t)(Lmax - Lmin);
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    const size_t total = (size_t)K * (size_t)M;
    // the K momenta are ONE run of `total` normals in memory; a pair may straddle two of them
    size_t first = 0;  // normals before the first pair: the value cached by an earlier call
    if (r->has_gauss && total > 0) {
        p0s[0] = r->gauss * sigma;
        r->has_gauss = 0;
        r->gauss = 0.0;
        first = 1;
    }
    const size_t npairs = (total - first + 1) / 2;
    const bool tail = (total - first) & 1;  // the last pair's second value has no slot: it is cached
    r->r2.resize(npairs);
    double *r2 = r->r2.data();
    double last[2] = {0.0, 0.0};
    size_t at = first, ip = 0;
    // The scale pass (sqrt(-2 log(r2) / r2) per accepted point: libm's log, as NumPy) runs BEHIND the generation on a
    // second thread, a trajectory's points at a time: the generation itself is sequential (MT19937's twists, the
    // rejections), ~2 of a draw's 3.5 ns per normal.  (Splitting the scale pass over several threads AFTER the
    // generation, as before, was slower than one thread on a GPU box's host: 4.3 against 3.5 ns per normal.)
    auto scale = [&](size_t p_lo, size_t p_hi) {
        for (size_t i = p_lo; i < p_hi; ++i) {
            const double f = rng_disc_factor(r2[i]);
            const size_t o = first + 2 * i;
            p0s[o] = (f * p0s[o]) * sigma;
            if (o + 1 < total) p0s[o + 1] = (f * p0s[o + 1]) * sigma;
        }
    };
    int nt = r->threads > 0 ? r->threads : env_int("GRAVHMC_RNG_THREADS", 2);
    const bool piped = nt > 1 && npairs >= 16384;
    std::atomic<size_t> produced{0};
    std::atomic<bool> finished{false};
    std::thread helper;
    if (piped)
        helper = std::thread([&]() {
            size_t lo = 0;
            for (;;) {
                const bool fin = finished.load(std::memory_order_acquire);
                const size_t hi = produced.load(std::memory_order_acquire);
                if (hi > lo) {
                    scale(lo, hi);
                    lo = hi;
                } else if (fin) {
                    break;
                } else {
                    std::this_thread::yield();
                }
            }
        });
    for (int k = 0; k < K; ++k) {
        // masked rejection on 32-bit outputs (none drawn when there is one possible value)
        uint32_t v = 0;
        if (rng != 0) {
            do {
                v = rng_next32(r) & mask;
            } while (v > rng);
        }
        L[k] = Lmin + (int)v;
        const size_t end = (size_t)(k + 1) * (size_t)M;  // pairs until this momentum is full
        if (at < end) {
            size_t np = (end - at + 1) / 2;
            const bool spill = tail && k == K - 1;
            if (spill) np -= 1;
            rng_disc_points(r, np, p0s + at, r2 + ip);
            at += 2 * np;
            ip += np;
            if (spill) {
                rng_disc_points(r, 1, last, r2 + ip);
                p0s[at] = last[0];
                at += 2;
                ip += 1;
            }
        }
        us[k] = rng_double(r);
        if (piped) produced.store(ip, std::memory_order_release);
    }
    if (piped) {
        finished.store(true, std::memory_order_release);
        helper.join();
    } else {
        scale(0, npairs);
    }
    if (tail) {
        r->gauss = rng_disc_factor(r2[npairs - 1]) * last[1];
        r->has_gauss = 1;
    }
    return GH_OK;
}
