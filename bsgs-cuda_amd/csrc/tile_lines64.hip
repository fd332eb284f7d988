// the tile kernels for 64-byte bucket lines (15 entries per bucket)
#define BSGS_TILE_MODE 2
#include "tile_launch.inc"
