// fpe_bits.hpp — bit-window helpers (placeholder; filled in with the bit-window kernels).
#pragma once
