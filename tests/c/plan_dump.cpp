// prints the chunk plan of a host-array call: ./plan_dump n unit in_bytes out_bytes ns_per_elem gens_fixed  ->  "off:m off:m ..."
#include <cstdio>
#include <cstdlib>
#include "pipeline_plan.h"
int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const auto plan = fq_plan::plan_pieces(strtoull(argv[1], 0, 10), strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10), strtoull(argv[4], 0, 10), atof(argv[5]), atoi(argv[6]));
    for (const auto& p : plan) printf("%zu:%zu ", p.off, p.m);
    printf("\n");
    return 0;
}
