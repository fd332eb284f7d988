// the tile kernels for 128-byte bucket lines (31 entries per bucket)
#define BSGS_TILE_MODE 3
#include "tile_launch.inc"
