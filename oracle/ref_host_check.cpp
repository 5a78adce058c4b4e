// ref_host_check.cpp -- a host program built ENTIRELY on the reference's own host code:
// its headers (include/AlignmentKernel.h, src/impl/CustomParameters.h, src/impl/CustomLogger.h)
// and its utility translation unit (src/util/versalignUtil.cpp: FastaProvider, pad(), DLL_init,
// DLL_function_retreival), compiled where they lie under $(REF) by oracle/Makefile.  It performs
// the reference main()'s example calls (src/impl/main.cpp:119-191) for a plugin given by path and
// writes the same four text files.  Test infrastructure: it proves that a host compiled against the
// reference's headers -- not this repo's restatement of them -- drives libHIPKernel.so unchanged.
// The binary lands in oracle/_ref/ (never committed); nothing of the reference is copied.
#include "AlignmentKernel.h"
#include "CustomParameters.h"
#include "CustomLogger.h"

#include "util/versalignUtil.h"

#include <fstream>
#include <iostream>
#include <string>

int main(int argc, char *argv[]) {
    if (argc < 5) {
        std::cerr << "usage: ref_host <plugin.so> <reads.fa> <refs.fa> <out-dir> [threads]" << std::endl;
        return 2;
    }
    const std::string plugin = argv[1], out_dir = argv[4];
    CustomParameters parameters;
    CustomLogger logger;

    FastaProvider provider;
    std::vector<const char *> reads_vec = provider.parse_fasta(argv[2]);
    std::vector<const char *> refs_vec = provider.parse_fasta(argv[3]);
    if (reads_vec.size() != refs_vec.size() || reads_vec.empty()) {
        std::cerr << "unequal or empty read / ref sets" << std::endl;
        return 1;
    }
    char const **const reads = &reads_vec[0];
    char const **const refs = &refs_vec[0];
    const size_t n = reads_vec.size();
    parameters.read_length = (int)pad(reads, (int)n, '\0');
    parameters.ref_length = (int)pad(refs, (int)n, '\0');
    parameters.num_threads = argc > 5 ? atoi(argv[5]) : 4;

    const int dll = DLL_init(plugin.c_str(), &parameters, &logger);
    if (dll < 0) return 1;
    fp_load_alignment_kernel spawn = (fp_load_alignment_kernel)DLL_function_retreival(dll, "spawn_alignment_kernel");
    fp_delete_alignment_kernel destroy = (fp_delete_alignment_kernel)DLL_function_retreival(dll, "delete_alignment_kernel");
    if (!spawn || !destroy) return 1;

    const char *const tags[2] = {"smith_waterman", "needleman_wunsch"};
    for (int mode = 0; mode < 2; ++mode) {
        AlignmentKernel *kernel = 0;
        try {
            kernel = spawn();
        } catch (const char *msg) {
            std::cerr << "spawn failed: " << msg << std::endl;
            return 1;
        }
        short *scores = new short[n]();
        Alignment *alignments = new Alignment[n]();
        kernel->score_alignments(mode, (int)n, reads, refs, scores);
        kernel->compute_alignments(mode, (int)n, reads, refs, alignments);
        std::ofstream out((out_dir + "/scores_" + tags[mode] + ".txt").c_str());
        for (size_t i = 0; i < n; ++i)      // the padded blocks carry no terminator: bound the text
            out << std::string(reads[i], strnlen(reads[i], parameters.read_length)) << "\t" << scores[i] << std::endl;
        out.close();
        out.open((out_dir + "/alignments_" + tags[mode] + ".txt").c_str());
        for (size_t i = 0; i < n; ++i) {
            out << alignments[i].read + alignments[i].readStart << std::endl;
            out << alignments[i].ref + alignments[i].refStart << std::endl << std::endl;
        }
        out.close();
        delete[] alignments;                // ~Alignment delete[]s the rows the plugin new[]'d
        delete[] scores;
        destroy(kernel);
    }
    return 0;
}
