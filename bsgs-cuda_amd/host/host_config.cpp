// host_config.cpp -- the reference's command line (getprogparam 1_9_7File.pb:875-1042), its -w / -htsz limits (4412-4472), the configuration fingerprint and the
// checkpoint file (saveCurentCNT 3897-3931).
#include "host.h"

// ---- SHA1 (configuration fingerprint of currentwork.txt, 1_9_7File.pb:4635-4636) -------------------------------
std::string sha1_hex(const std::string &msg)
{
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    std::string m = msg;
    const uint64_t bits = (uint64_t)msg.size() * 8;
    m.push_back((char)0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 7; i >= 0; i--) m.push_back((char)(bits >> (8 * i)));
    auto rol = [](uint32_t v, int s) { return (v << s) | (v >> (32 - s)); };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[80];
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)(uint8_t)m[off + 4 * i] << 24) | ((uint32_t)(uint8_t)m[off + 4 * i + 1] << 16) |
                   ((uint32_t)(uint8_t)m[off + 4 * i + 2] << 8) | (uint32_t)(uint8_t)m[off + 4 * i + 3];
        for (int i = 16; i < 80; i++) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; i++) {
            uint32_t f, k;
            if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
            else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
            else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
            const uint32_t t = rol(a, 5) + f + e + k + w[i];
            e = d; d = c; c = rol(b, 30); b = a; a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
    }
    char out[41];
    snprintf(out, sizeof out, "%08x%08x%08x%08x%08x", h[0], h[1], h[2], h[3], h[4]);
    return out;
}

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(1);
}
std::string cut_hex(std::string s)
{
    if (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) s = s.substr(2);
    for (auto &c : s) c = (char)tolower(c);
    return s;
}

void usage(const Config &c)
{
    printf(" -t      Number of GPU threads, default %u\n -b      Number of GPU blocks, default %u\n -p      Number of pparam, default %u\n"
           " -d      Select GPU IDs, default all\n-pb      Set single uncompressed/compressed pubkey for searching\n"
           "-pk      Range start from , default %s\n-pke     End range \n-w       Set number of baby items 2^ or decimal representation\n"
           "-htsz    Set number of HashTable 2^ , default %u\n-infile  Set file with pubkey for searching in uncompressed/compressed  format (search sequential)\n"
           "-wl      Set recovery file from which the state will be loaded\n-wt      Set timer for autosaving current state, default every %dseconds\n"
           "-onlygen Generate the table files and exit (onlygen_1_9_6File0.exe)\n-cpugen  Build missing table / giants files on the host CPU; with -onlygen no GPU is touched (the reference`s CPU-only generator)\n-dir     Directory for table files, currentwork.txt and win.txt\n"
           "-ext     Extended baby table built in GPU memory (no HT files); automatic for -w above the reference limit, up to 2^36\n"
           "-noverify    Skip the verification before the search (census of the table, sampled k*G and giants on every GPU; several GPUs: replicas compared)\n"
           "-refquirks   Reproduce the reference kernel's -Gy borrow bug bit for bit (default: correct arithmetic, finds a superset)\n"
           "-hostcentres Add the tile centres on the host and upload them (default: derived on the GPU from the tile counter)\n"
           "-tune        Time a few placements of the GPU buffers at start-up and keep the fastest (the engine already places them by grade)\n"
           "-startup     Several GPUs: broadcast (GPU 0 holds the table, the others receive it over xGMI), local (every GPU builds / uploads its own),\n"
           "             allgather (extended tables: every GPU builds 1/N of the bucket lines, then all-gather); default: local for extended tables, else broadcast\n"
           "-transport   Several GPUs: rccl | peer (direct peer copies) | auto (RCCL when the GPUs are distinct and librccl loads)\n"
           "-lanes       -infile: public keys searched side by side, each on an engine of its own per GPU (default: 2 when a job is only a launch or two long, else 1)\n"
           "-w auto      The table Tune picks for the range given: the one that minimises table build + worst-case search (a 64-bit range: -w 30 -ext)\n"
           "-buckets     Extended table: the number of buckets itself (any number below 2^32; 64-byte lines up to 12.5 items per bucket, else 128-byte lines), e.g. -w 35 -buckets 3221225472\n",
           c.t, c.b, c.p, c.pk.c_str(), c.htsz, c.wt);
}

Config parse_args(int argc, char **argv)
{
    Config c;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        for (auto &ch : a) ch = (char)tolower(ch);
        auto next = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "-h") { usage(c); exit(0); }
        else if (a == "-t") { c.t = (uint32_t)atoi(next().c_str()); printf("Number of GPU threads set to #%u\n", c.t); }
        else if (a == "-b") { c.b = (uint32_t)atoi(next().c_str()); printf("Number of GPU blocks set to #%u\n", c.b); }
        else if (a == "-p") { c.p = (uint32_t)atoi(next().c_str()); printf("Number of pparam set to #%u\n", c.p); }
        else if (a == "-d") { c.devices = next(); printf("Used GPU devices #%s\n", c.devices.c_str()); }
        else if (a == "-pb") { c.pub = cut_hex(next()); printf("Pubkey set to %s\n", c.pub.c_str()); }
        else if (a == "-pk") { c.pk = cut_hex(next()); printf("Range begin: 0x%s\n", c.pk.c_str()); }
        else if (a == "-pke") { c.pke = cut_hex(next()); c.pke_given = true; printf("Range end: 0x%s\n", c.pke.c_str()); }
        else if (a == "-w") {                                   // <=32: 2^value (fractional allowed), else decimal  (1009-1022)
            const std::string v = next();
            if (v == "auto" || v == "AUTO") { c.w_auto = true; printf("Items number: chosen for the range (Tune)\n"); continue; }
            const double d = atof(v.c_str());
            // the reference switches to decimal above 32; 33..36 are exponents of the extended table here
            if (d <= 36.0) { c.w = (uint64_t)std::pow(2.0, d); printf("Items number set to 2^%s=%llu\n", v.c_str(), (unsigned long long)c.w); }
            else { c.w = strtoull(v.c_str(), nullptr, 10); printf("Items number set to %llu = 2^%f\n", (unsigned long long)c.w, std::log2((double)c.w)); }
        }
        else if (a == "-htsz") {
            const std::string v = next();
            const double d = atof(v.c_str());
            c.htsz = (uint32_t)d; c.htsz_arg = c.htsz;
            if (d != std::floor(d)) {               // a fraction (as -w takes one, 1_9_7File.pb:1009-1022): extended tables may have any number of buckets
                c.htsz_arg = (uint32_t)std::llround(std::pow(2.0, d));
                printf("HT size set to 2^%s=%u buckets (extended table)\n", v.c_str(), c.htsz_arg);
            } else printf("HT size set to 2^%u\n", c.htsz);
        }
        else if (a == "-buckets") { c.htsz_arg = (uint32_t)strtoull(next().c_str(), nullptr, 10); c.htsz = 0; while ((2ull << c.htsz) <= c.htsz_arg) c.htsz++; printf("HT size set to %u buckets (extended table)\n", c.htsz_arg); }
        else if (a == "-sf") { c.file_search = atoi(next().c_str()) != 0; printf(c.file_search ? "Search in file\n" : "Search in RAM\n"); }
        else if (a == "-lanes") { c.lanes = atoi(next().c_str()); if (c.lanes < 1 || c.lanes > 4) die("-lanes 1..4"); }
        else if (a == "-startup") { c.startup = next(); for (auto &ch : c.startup) ch = (char)tolower(ch); }
        else if (a == "-transport") { c.transport = next(); for (auto &ch : c.transport) ch = (char)tolower(ch); }
        else if (a == "-infile") { c.infile = next(); printf("Will be used file: %s\n", c.infile.c_str()); }
        else if (a == "-wl") { c.recovery_file = next(); printf("Recovery work file: %s\n", c.recovery_file.c_str()); }
        else if (a == "-wt") { c.wt = std::max(30, atoi(next().c_str())); printf("Saving timer every %d seconds\n", c.wt); }
        else if (a == "-onlygen") c.onlygen = true;
        else if (a == "-cpugen") c.cpugen = true;
        else if (a == "-maxtiles") c.max_tiles = strtoull(next().c_str(), nullptr, 10);
        else if (a == "-dir") c.dir = next();
        else if (a == "-ext") c.ext = true;
        else if (a == "-refquirks") c.ref_quirks = true;
        else if (a == "-verifyreplicas") c.verify_replicas = true;
        else if (a == "-noverify") c.verify_replicas = false;
        else if (a == "-hostcentres") c.host_centres = true;
        else if (a == "-tune") c.tune = true;
        else if (a == "-joblog") c.joblog = next();
        else die("Unknown parameter " + a);
    }
    // limits 1_9_7File.pb:4412-4418, 4616-4618
    if (c.w >= 3069485951ull) {                    // beyond the reference's u32 file format: extended device-resident table
        if (c.w > (1ull << 36)) die("-w must be at most 2^36");
        c.ext = true;
        printf("-w above the reference limit 3069485951: extended table in GPU memory, no HT files\n");
    }
    if (c.htsz > 31 || c.htsz < 1) die("-htsz must be 1..31");
    if (c.htsz_arg > 31) {
        if (!(c.htsz_arg & (c.htsz_arg - 1))) { c.htsz = 0; while ((1u << c.htsz) < c.htsz_arg) c.htsz++; c.htsz_arg = c.htsz; }      // a power of two after all
        else { c.ext = true; printf("%u buckets (not a power of two): extended table in GPU memory, no HT files\n", c.htsz_arg); }
    }
    if (c.startup != "auto" && c.startup != "broadcast" && c.startup != "local" && c.startup != "allgather") die("-startup: broadcast | local | allgather | auto");
    if (c.transport != "auto" && c.transport != "rccl" && c.transport != "peer") die("-transport: rccl | peer | auto");
    if (c.p & 1) die("-p must be even");
    if (!c.t || !c.b || !c.p) die("-t -b -p must be non-zero");
    return c;
}

// ---- the reference's limits on -w / -htsz for tables in ITS format (1_9_7File.pb:4412-4472): -w below 3069485951, -htsz below 32, and the "UNSAFE mode" question
// (answer Y on stdin to go on) where duplicate 32-bit values in one bucket become likely; then its warning about a -htsz that is too low.  Extended tables
// (built in GPU memory, no HT files: -ext, -w above 2^32, -buckets) are outside that format and outside these limits.  Returns "" to go on, else the exit message.
std::string table_limits(uint64_t w, uint32_t htsz, FILE *answers)
{
    if (w >= 3069485951ull) return "-w should be less or equil to 3069485951 Or 2^31.515349920643907";
    if (htsz > 31) return "-htsz should be less than 32";
    static const struct { uint32_t htsz; uint64_t limit; const char *shown; } unsafe[] = {
        {27, 1331331443ull, "1331331443 or 2^30.310222637591963"}, {28, 1777178603ull, "1777178603 Or 2^30.726941530690112"}, {29, 3069485951ull, "3069485950 Or 2^31.515349920643907"},
        {30, 3069485951ull, "3069485951 Or 2^31.515349920643907"}, {31, 3069485951ull, "3069485951 Or 2^31.515349920643907"}};
    for (const auto &u : unsafe)
        if (htsz == u.htsz && w > u.limit) {
            printf("With -htsz %u value -w should be less or equil to %s\nDue to the possibility of duplicate values in the hash table\n"
                   "It is unsafe to use values higher than those specified above\nTo continue in UNSAFE mode type Y and press ENTER\n", u.htsz, u.shown);
            fflush(stdout);
            char line[64] = {0};
            if (!answers || !fgets(line, sizeof line, answers)) return " ";
            std::string ans(line);
            while (!ans.empty() && (ans.back() == '\n' || ans.back() == '\r')) ans.pop_back();
            if (ans != "Y") return " ";
        }
    const int need = (int)std::floor(std::log2((double)w)) - (int)htsz;
    if (need > 3) printf("WARNING! -htsz parametr is to low, should be at least %d\n", (int)std::floor(std::log2((double)w)) - 2);
    return "";
}

// ---- checkpoint: saveCurentCNT 1_9_7File.pb:3897-3931 ------------------------------------------------------------
std::string fingerprint(const Config &c)
{
    std::ostringstream s;
    s << c.t << c.b << c.p << c.w << c.pk << c.pke << (c.htsz_arg > 31 ? c.htsz_arg : c.htsz);     // Str(t)+Str(b)+Str(p)+Str(w)+pk+pke+Str(htsz)  (4635-4636); a bucket count stands for htsz
    return sha1_hex(s.str());
}
void save_checkpoint(Shared &S)
{
    // the minimum counter over the GPUs' unfinished batches (a restart re-does at most the batches in flight); both locks are
    // held so that a batch cannot leave the dispenser between reading its counter and reading the in-flight table
    Scalar cnt;
    {
        std::lock_guard<std::mutex> lk(S.job_mutex);
        std::lock_guard<std::mutex> lk2(S.inflight_mutex);
        cnt = S.glob_key;
        for (size_t g = 0; g < S.inflight.size(); g++) if (S.inflight_valid[g] && hs::fe_cmp(S.inflight[g], cnt) < 0) cnt = S.inflight[g];
        if (S.joblog) { fprintf(S.joblog, "save %s\n", hs::fe_to_hex(cnt).c_str()); fflush(S.joblog); }
    }
    static std::mutex file_mutex;                                   // several lanes (and a job that has just ended) may save at the same moment
    std::lock_guard<std::mutex> fl(file_mutex);
    const std::string tmp = S.cfg.dir + "/currentwork.temp", dst = S.cfg.dir + "/currentwork.txt";
    {
        std::ofstream f(tmp, std::ios::binary);
        f << S.listpos << "\r\n" << S.mainpub_hex << "\r\n" << hs::fe_to_hex(cnt) << "\r\n" << fingerprint(S.cfg) << "\r\n";
    }
    rename(tmp.c_str(), dst.c_str());
}
