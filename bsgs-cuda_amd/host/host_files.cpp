// host_files.cpp -- the table files on the host: buffers, readers / writers (Save_HTpacked 1_9_7File.pb:3645-3759, Save_Load_Giants 1905-2058) and the CPU-only generator
// (-cpugen).
#include "host.h"

// Large images (the 5.4 + 9.6 GB of a -w 30 table) are taken 2 MiB-aligned, offered to transparent huge pages and FIRST-TOUCHED BY SEVERAL THREADS: the kernel
// clears every page it hands out, and one thread faulting 15 GB in (inside a device-to-host copy or a read()) is most of a 3.4 s "build + bring to the host" stage
void HostBuf::resize(uint64_t bytes)
{
    free(p); p = nullptr; n = 0;
    if (!bytes) return;
    const bool big = bytes >= (256ull << 20) && !getenv("BSGS_HOST_NO_PREFAULT");
    p = big ? (uint8_t *)aligned_alloc(2u << 20, (bytes + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1)) : (uint8_t *)malloc(bytes);
    if (!p) { fprintf(stderr, "out of host memory (%llu bytes)\n", (unsigned long long)bytes); exit(1); }
    n = bytes;
    if (!big) return;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
    const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([this, bytes, q, nth]() {
        const uint64_t lo = bytes / nth * q, hi = q + 1 == nth ? bytes : bytes / nth * (q + 1);
        for (uint64_t o = lo; o < hi; o += 4096) ((volatile uint8_t *)p)[o] = 0;
    });
    for (auto &t : th) t.join();
}

bool file_has_size(const std::string &path, uint64_t expect)
{
    struct stat st;
    return stat(path.c_str(), &st) == 0 && (uint64_t)st.st_size == expect;
}
bool read_file(const std::string &path, HostBuf &out, uint64_t expect)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const uint64_t n = (uint64_t)f.tellg();
    if (n != expect) return false;
    out.resize(n);
    f.seekg(0);
    f.read((char *)out.data(), (std::streamsize)n);
    return (bool)f;
}
// written under a temporary name and renamed after a checked flush: a run that ends while the file is being written (a later start-up error, a full disk)
// leaves a `.part` file behind, never a short table under the reference's name
void write_file(const std::string &path, const void *p, uint64_t n)
{
    const std::string tmp = path + ".part";
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) die("Can`t create " + tmp);
        f.write((const char *)p, (std::streamsize)n);
        f.flush();
        if (!f) { remove(tmp.c_str()); die("Can`t write " + path + " (" + std::to_string(n) + " bytes): disk full?"); }
    }
    if (rename(tmp.c_str(), path.c_str()) != 0) die("Can`t rename " + tmp);
}

// ---- -cpugen: the table and giants files built on the HOST CPU (BASELINE config 1 as it is worded; the reference's CPU-only generator is a program of its own,
// onlygen1_9_6File.pb:2915-3204, over lib/Curve64.pb).  Plumbing, not a fast path: k*G for k = 1..w by affine additions with batched normalisation (host_secp.h), one
// range of k per host thread; entries filed by bucket (counting sort), each bucket ascending by (hash, position) -- the order of the reference's sorted buckets
// (1_9_7File.pb:2771-2820) and of the GPU builder; images as in SURVEY.md Appendix C (1_9_7File.pb:3232-3444).  Byte-identical to the GPU builder's files (CPU test).
void cpu_build_tables(uint64_t w, uint32_t htsz, uint8_t *htgpu, uint8_t *htcpu)
{
    const uint64_t items = 1ull << htsz;
    std::vector<uint64_t> key(w);                                     // low 64 bits of x(k*G) at index k - 1
    const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max(1u, std::thread::hardware_concurrency()), 64ull, (w + 65535) / 65536}));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([&, q]() {
        const uint64_t lo = w * q / nth, hi = w * (q + 1) / nth;      // k - 1 in [lo, hi)
        for (uint64_t first = lo; first < hi; first += 65536) {
            const size_t cnt = (size_t)std::min<uint64_t>(65536, hi - first);
            const std::vector<Affine> pts = hs::strided_multiples(hs::G, first + 1, 1, cnt);
            for (size_t i = 0; i < cnt; i++) key[first + i] = pts[i].x.l[0];
        }
    });
    for (auto &t : th) t.join();
    std::vector<uint32_t> off(items + 1, 0);
    for (uint64_t i = 0; i < w; i++) off[((uint32_t)key[i] & (uint32_t)(items - 1)) + 1]++;
    for (uint64_t b = 0; b < items; b++) off[b + 1] += off[b];       // off[b] = entries in buckets below b
    std::vector<uint64_t> ent(w);                                     // hash << 32 | position: ascending = (hash, position)
    {
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        for (uint64_t i = 0; i < w; i++) ent[cur[(uint32_t)key[i] & (uint32_t)(items - 1)]++] = (key[i] >> 32 << 32) | i;
    }
    for (uint64_t b = 0; b < items; b++) std::sort(ent.begin() + off[b], ent.begin() + off[b + 1]);
    uint32_t *g = (uint32_t *)htgpu, *c = (uint32_t *)htcpu;
    memcpy(g, off.data(), 4 * (items + 1));                           // starts, then the total (= w)
    memcpy(c, off.data(), 4 * (items + 1));
    for (uint64_t i = 0; i < w; i++) {
        g[items + 1 + i] = (uint32_t)(ent[i] >> 32);
        c[items + 1 + 2 * i] = (uint32_t)(ent[i] >> 32);
        c[items + 1 + 2 * i + 1] = (uint32_t)ent[i];
    }
}
// G2[i] = (i + 1) * A, i < t*b*p, in the strided file layout (1_9_7File.pb:1831-1903, 1954-1970): the k-th MOST significant 32-bit word of coordinate c of G2[i]
// at u32 index c*8*maxnonce + ((i % p)*8 + k)*T + i / p, T = t*b
void cpu_build_g2(const Affine &A, uint32_t t, uint32_t b, uint32_t p, uint8_t *g2)
{
    const uint64_t T = (uint64_t)t * b, maxnonce = T * p;
    uint32_t *out = (uint32_t *)g2;
    const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max(1u, std::thread::hardware_concurrency()), 64ull, (maxnonce + 65535) / 65536}));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([&, q]() {
        const uint64_t lo = maxnonce * q / nth, hi = maxnonce * (q + 1) / nth;
        for (uint64_t first = lo; first < hi; first += 65536) {
            const size_t cnt = (size_t)std::min<uint64_t>(65536, hi - first);
            const std::vector<Affine> pts = hs::strided_multiples(A, first + 1, 1, cnt);
            for (size_t j = 0; j < cnt; j++) {
                const uint64_t i = first + j;
                for (int c = 0; c < 2; c++) {
                    const hs::Fe &v = c ? pts[j].y : pts[j].x;
                    for (int k = 0; k < 8; k++) out[(uint64_t)c * 8 * maxnonce + ((i % p) * 8 + k) * T + i / p] = (uint32_t)(v.l[3 - k / 2] >> (32 * (1 - k % 2)));
                }
            }
        }
    });
    for (auto &x : th) x.join();
}
