// valign-bench -- command-line counterpart of the reference host program.
//
// The reference's src/impl/main.cpp loads ../testset/{reads,refs}.fa, pads them, runs one
// hard-coded kernel in Smith-Waterman and Needleman-Wunsch mode, writes four text files and
// prints a thread-ladder timing table (main.cpp:85-110, 129-191, 197-212, 240-295).  This tool
// does the same for ANY versalignLib plugin given by path, through the same dlopen protocol
// (libvalignhost.so), so outputs of the reference's CPU kernels and of libHIPKernel.so can be
// diffed textually.
//
//   valign-bench --kernel <plugin.so> --reads reads.fa --refs refs.fa [--out-dir DIR]
//                [--threads 10] [--ladder 1,2,4,8] [--loops 100] [--time score|align|none]
//                [--param key=value ...] [--cigar]
//
// Output files (same names and line formats as the reference):
//   scores_smith_waterman.txt / scores_needleman_wunsch.txt ....... "<read>\t<score>"
//   alignments_smith_waterman.txt / alignments_needleman_wunsch.txt  "<read row>\n<ref row>\n\n"
// With --cigar also cigars_smith_waterman.txt / cigars_needleman_wunsch.txt: "<read>\t<CIGAR>" (=/X/I/D;
// not a reference format -- the reference only prints the gapped rows).
// Timing table on stdout: "Threads\t<t1>\t<t2>...\n<kernel>\t<us per call>..." (main.cpp:197-212,292).
#include "valign_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

namespace {

struct Options {
    std::string kernel, reads, refs, out_dir = ".", time_mode = "align";
    int threads = 10, loops = 100;
    bool cigar = false;
    std::vector<int> ladder = {1, 2, 4, 8, 16, 32, 64};
    std::vector<std::pair<std::string, int>> params;
};

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "valign-bench: %s\n", msg.c_str());
    exit(2);
}

Options parse(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) die("missing value after " + a);
            return argv[++i];
        };
        if (a == "--kernel") o.kernel = next();
        else if (a == "--reads") o.reads = next();
        else if (a == "--refs") o.refs = next();
        else if (a == "--out-dir") o.out_dir = next();
        else if (a == "--threads") o.threads = atoi(next().c_str());
        else if (a == "--loops") o.loops = atoi(next().c_str());
        else if (a == "--time") o.time_mode = next();
        else if (a == "--cigar") o.cigar = true;
        else if (a == "--ladder") {
            o.ladder.clear();
            std::string v = next();
            size_t pos = 0;
            while (pos < v.size()) {
                size_t comma = v.find(',', pos);
                if (comma == std::string::npos) comma = v.size();
                o.ladder.push_back(atoi(v.substr(pos, comma - pos).c_str()));
                pos = comma + 1;
            }
        } else if (a == "--param") {
            std::string v = next();
            size_t eq = v.find('=');
            if (eq == std::string::npos) die("--param wants key=value");
            o.params.emplace_back(v.substr(0, eq), atoi(v.substr(eq + 1).c_str()));
        } else {
            die("unknown argument " + a);
        }
    }
    if (o.kernel.empty() || o.reads.empty() || o.refs.empty()) die("--kernel, --reads and --refs are required");
    return o;
}

struct Batch {
    uint8_t *data = nullptr;
    int count = 0, length = 0;
};

Batch load(const std::string &path) {
    char *blob = nullptr;
    int count = 0;
    if (vh_parse_fasta(path.c_str(), &blob, &count) != 0) die(vh_last_error());
    Batch b;
    b.count = count;
    if (vh_pad(blob, count, '\0', &b.data, &b.length) != 0) die(vh_last_error());
    vh_free(blob);
    return b;
}

vh_plugin *spawn(const Options &o, int R, int F, int threads) {
    vh_plugin *p = vh_open(o.kernel.c_str());
    if (!p) die(vh_last_error());
    vh_log_to_stderr(p, 1);
    vh_set_param(p, "read_length", R);
    vh_set_param(p, "ref_length", F);
    vh_set_param(p, "num_threads", threads);
    for (auto &kv : o.params) vh_set_param(p, kv.first.c_str(), kv.second);
    if (vh_spawn(p) != 0) die(vh_last_error());
    return p;
}

// the reference prints `char *` sequences with operator<<: text up to the first NUL
std::string c_text(const uint8_t *s, int max_len) {
    int n = 0;
    while (n < max_len && s[n] != 0) ++n;
    return std::string((const char *)s, (size_t)n);
}

void run_mode(vh_plugin *p, int opt, const char *tag, const Batch &reads, const Batch &refs, const std::string &dir,
              bool cigar) {
    const int n = reads.count, AL = reads.length + refs.length;
    std::vector<int16_t> scores((size_t)n, 0);
    if (vh_score(p, opt, n, reads.data, refs.data, scores.data()) != 0) die(vh_last_error());
    FILE *f = fopen((dir + "/scores_" + tag + ".txt").c_str(), "w");
    if (!f) die("cannot write into " + dir);
    for (int i = 0; i < n; ++i)
        fprintf(f, "%s\t%d\n", c_text(reads.data + (size_t)i * reads.length, reads.length).c_str(), (int)scores[i]);
    fclose(f);
    std::vector<uint8_t> rows((size_t)n * 2 * AL);
    std::vector<int16_t> idx((size_t)n * 4);
    if (vh_align(p, opt, n, reads.data, refs.data, rows.data(), idx.data(), 1) != 0) die(vh_last_error());
    f = fopen((dir + "/alignments_" + tag + ".txt").c_str(), "w");
    if (!f) die("cannot write into " + dir);
    for (int i = 0; i < n; ++i) {
        const uint8_t *rr = rows.data() + (size_t)i * 2 * AL, *fr = rr + AL;
        const int rs = idx[4 * i], fs = idx[4 * i + 2];
        fprintf(f, "%s\n%s\n\n", c_text(rr + rs, AL - rs).c_str(), c_text(fr + fs, AL - fs).c_str());
    }
    fclose(f);
    if (!cigar) return;
    f = fopen((dir + "/cigars_" + tag + ".txt").c_str(), "w");
    if (!f) die("cannot write into " + dir);
    std::vector<char> text((size_t)4 * AL + 16);
    for (int i = 0; i < n; ++i) {
        const uint8_t *rr = rows.data() + (size_t)i * 2 * AL;
        if (vh_cigar(rr, rr + AL, idx[4 * i], idx[4 * i + 1], 1, text.data(), (int)text.size()) < 0) die(vh_last_error());
        fprintf(f, "%s\t%s\n", c_text(reads.data + (size_t)i * reads.length, reads.length).c_str(), text.data());
    }
    fclose(f);
}

}  // namespace

int main(int argc, char **argv) {
    const Options o = parse(argc, argv);
    const Batch reads = load(o.reads), refs = load(o.refs);
    if (reads.count != refs.count) {
        fprintf(stderr, "DRASTIC\t[MAIN]\tUnequal sizes of reads and ref set (%d vs %d).\n", reads.count, refs.count);
        return -1;
    }
    vh_plugin *p = spawn(o, reads.length, refs.length, o.threads);
    run_mode(p, 0, "smith_waterman", reads, refs, o.out_dir, o.cigar);
    vh_close(p);
    p = spawn(o, reads.length, refs.length, o.threads);     // the reference respawns per mode
    run_mode(p, 1, "needleman_wunsch", reads, refs, o.out_dir, o.cigar);
    vh_close(p);

    if (o.time_mode != "none") {
        const int n = reads.count, AL = reads.length + refs.length;
        printf("Threads");
        for (int t : o.ladder) printf("\t%d", t);
        printf("\n%s", o.kernel.c_str());
        std::vector<int16_t> scores((size_t)n);
        std::vector<uint8_t> rows((size_t)n * 2 * AL);
        std::vector<int16_t> idx((size_t)n * 4);
        for (int t : o.ladder) {
            vh_plugin *k = spawn(o, reads.length, refs.length, t);
            vh_log_to_stderr(k, 0);
            auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < o.loops; ++rep) {
                const int rc = o.time_mode == "score"
                                   ? vh_score(k, 0, n, reads.data, refs.data, scores.data())
                                   : vh_align(k, 0, n, reads.data, refs.data, rows.data(), idx.data(), 0);
                if (rc != 0) die(vh_last_error());
            }
            auto t1 = std::chrono::steady_clock::now();
            printf("\t%.0f", std::chrono::duration<double, std::micro>(t1 - t0).count() / o.loops);
            fflush(stdout);
            vh_close(k);
        }
        printf("\n");
    }
    vh_free(reads.data);
    vh_free(refs.data);
    return 0;
}
