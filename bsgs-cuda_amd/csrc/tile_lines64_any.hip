// the tile kernels for 64-byte bucket lines of an extended table with ANY number of buckets (the bucket from 48 bits of the key: giant_kernel.hip.h bucket_mul48).
// Their own instantiation: the kernels of power-of-two tables (tile_lines64.hip) must not even read TileArgs::bucket_mul.
#define BSGS_TILE_MODE 4
#include "tile_launch.inc"
