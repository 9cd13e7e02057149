// Translation unit 3 of 5 of the generalised register-resident kernels (tile_gen.inc): split so that the instantiations build in parallel.
#define GEN_PART 3
#include "tile_gen.inc"
