// Translation unit 1 of 5 of the generalised register-resident kernels (tile_gen.inc): split so that the instantiations build in parallel.
#define GEN_PART 1
#include "tile_gen.inc"
