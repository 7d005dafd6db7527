// C++ caller of the host mirror: graph directory + BAM file in, the reference's result files out, the way HLA-LA.cpp drives
// processBAM::alignReads and HLATyper::HLATypeInference (HLA-LA.cpp:760-900).  The Python test writes the input files, runs this
// program and compares its files with the ones the ctypes path wrote.
#include <cstdio>
#include <string>

#include "../../hla-la_amd/host/hlala_host.hpp"

using namespace hlala::host;

int main(int argc, char** argv)
{
    if(argc < 5) { std::printf("usage: %s <graphDir> <BAM> <outputDirectory> <locus>[,<locus>...] [hla_nom_g.txt]\n", argv[0]); return 64; }
    try {
        mapper::processBAM pB(argv[1], /*extendedReferenceGenome=*/false, 384, /*rng_seed=*/5);
        pB.alignReads(argv[2]);
        hla::HLATyper typer(argv[1], argc > 5 ? argv[5] : "");
        typer.filterParams.first20_n = 6;                                   // low coverage in the test sample
        std::vector<std::string> loci; std::string l = argv[4];
        for(size_t p = 0;;) { size_t q = l.find(',', p); loci.push_back(l.substr(p, q == std::string::npos ? q : q - p)); if(q == std::string::npos) break; p = q + 1; }
        for(const hla::HLATyper::bestGuess& g : typer.HLATypeInference(pB, argv[3], loci))
            std::printf("TYPED %s %s %s coverage %.3f\n", g.locus.c_str(), g.allele1.c_str(), g.allele2.c_str(), g.summary.locus_coverage);
        std::printf("insert size %.3f %.3f pairs %d\n", pB.IS_mean, pB.IS_sd, pB.n_units);
    } catch(std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    return 0;
}
