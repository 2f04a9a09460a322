// prints the chunk plan of a host-array call: ./plan_dump n unit in_bytes out_bytes ns_per_elem gens_fixed [link_in link_out]  ->  "off:m off:m ..."
#include <cstdio>
#include <cstdlib>
#include "pipeline_plan.h"
int main(int argc, char** argv) {
    if (argc != 7 && argc != 9) return 2;
    fq_plan::Rates r{ atof(argv[5]), fq_plan::LINK_BYTES_PER_NS, fq_plan::LINK_BYTES_PER_NS };
    if (argc == 9) { r.link_in = atof(argv[7]); r.link_out = atof(argv[8]); }
    const auto plan = fq_plan::plan_pieces(strtoull(argv[1], 0, 10), strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10), strtoull(argv[4], 0, 10), r, atoi(argv[6]));
    for (const auto& p : plan) printf("%zu:%zu ", p.off, p.m);
    printf("\n");
    return 0;
}
