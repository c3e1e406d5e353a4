// KeyMatchFull -- command-line front end of the GPU matcher, same process boundary as the reference tool
// (/root/reference/src/KeyMatchFull.cpp:59-76: `KeyMatchFull <list.txt> <outfile> [window_radius]`; the list holds one
// key-file path per line; a missing `x.key` is retried as `x.key.gz`, keys2a.cpp:87-110; Lowe's ASCII key format,
// keys2a.cpp:183-190; ratio 0.6; output format KeyMatchFull.cpp:131-142, read back by BundleIO.cpp:112-166).
// The matching itself is bsfm_key_match_full (C-ABI, include/bsfm.h): exact 2-NN on the MI355X, no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <zlib.h>
#include "bsfm.h"

namespace {

bool slurp_plain(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, got);
    fclose(f);
    return true;
}

bool slurp_gzip(const std::string& path, std::string& out)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) return false;
    char buf[1 << 16];
    int got;
    while ((got = gzread(g, buf, sizeof buf)) > 0) out.append(buf, (size_t)got);
    gzclose(g);
    return true;
}

// Whitespace-separated numbers; the descriptor entries are parsed as integers, the four location fields are skipped.
struct Cursor {
    const char* p; const char* end;
    void skip_ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool skip_token() { skip_ws(); if (p >= end) return false; while (p < end && !(*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; return true; }
    bool next_int(long& v)
    {
        skip_ws();
        if (p >= end) return false;
        bool neg = false;
        if (*p == '-') { neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        long x = 0;
        while (p < end && *p >= '0' && *p <= '9') x = 10 * x + (*p++ - '0');
        v = neg ? -x : x;
        return true;
    }
};

// Returns the number of keys (0 on any problem, like the reference reader) and fills desc with num x 128 bytes.
int read_key_file(const std::string& path, std::vector<unsigned char>& desc)
{
    std::string text;
    if (!slurp_plain(path, text) && !slurp_gzip(path + ".gz", text)) {
        printf("Could not open file: %s\n", path.c_str());
        return 0;
    }
    Cursor c{ text.data(), text.data() + text.size() };
    long num = 0, len = 0;
    if (!c.next_int(num) || !c.next_int(len) || num < 0) { printf("Invalid keypoint file\n"); return 0; }
    if (len != 128) { printf("Keypoint descriptor length invalid (should be 128)."); return 0; }
    desc.resize((size_t)num * 128);
    for (long i = 0; i < num; ++i) {
        for (int q = 0; q < 4; ++q)
            if (!c.skip_token()) { printf("Invalid keypoint file format."); desc.clear(); return 0; }
        for (int q = 0; q < 128; ++q) {
            long v;
            if (!c.next_int(v)) { printf("Invalid keypoint file format."); desc.clear(); return 0; }
            desc[(size_t)i * 128 + q] = (unsigned char)v;      // %hhu semantics: modulo 256
        }
    }
    return (int)num;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3 && argc != 4) {
        printf("Usage: %s <list.txt> <outfile> [window_radius]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const char* list_in = argv[1];
    const char* file_out = argv[2];
    const int window_radius = argc == 4 ? atoi(argv[3]) : -1;
    if (bsfm_device_count() <= 0) {
        fprintf(stderr, "[KeyMatchFull] no HIP device: this build has no CPU matcher\n");
        return EXIT_FAILURE;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::string> files;
    {
        std::string text;
        if (!slurp_plain(list_in, text)) { printf("Error opening file %s for reading\n", list_in); return EXIT_FAILURE; }
        size_t pos = 0;
        while (pos < text.size()) {
            size_t e = text.find('\n', pos);
            if (e == std::string::npos) e = text.size();
            std::string line = text.substr(pos, e - pos);
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (!line.empty()) files.push_back(line);
            pos = e + 1;
        }
    }
    const int num_images = (int)files.size();
    std::vector<std::vector<unsigned char>> store(num_images);
    std::vector<int> num_keys(num_images);
    std::vector<const unsigned char*> keys(num_images);
    for (int i = 0; i < num_images; ++i) {
        num_keys[i] = read_key_file(files[i], store[i]);
        keys[i] = store[i].empty() ? nullptr : store[i].data();
    }
    const auto t1 = std::chrono::steady_clock::now();
    printf("[KeyMatchFull] Reading keys took %0.3fs\n", std::chrono::duration<double>(t1 - t0).count());
    const int rc = bsfm_key_match_full(num_images, num_keys.data(), keys.data(), 0.6, window_radius, file_out);
    const auto t2 = std::chrono::steady_clock::now();
    printf("[KeyMatchFull] Matching took %0.3fs\n", std::chrono::duration<double>(t2 - t1).count());
    return rc < 0 ? EXIT_FAILURE : EXIT_SUCCESS;
}
