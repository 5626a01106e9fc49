// TEST INFRASTRUCTURE: stands in for the HIP runtime header when the kernel sources are compiled for the SIMT emulator.
#include "../../simt.h"
