// stand-in for the CUDA-side header of this name: everything lives in cuemu.h (test infrastructure)
#include "cuemu.h"
#include "mxshim.h"
