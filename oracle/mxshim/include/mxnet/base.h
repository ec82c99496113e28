// stand-in for the MXNet header of this name: everything lives in mxshim.h (test infrastructure)
#include "mxshim.h"
