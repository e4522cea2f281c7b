// TEST INFRASTRUCTURE (oracle/_ref): stands in for the CUDA-toolkit header of this name; everything lives in cuda_on_host.h.
#pragma once
#include <cuda_on_host.h>
