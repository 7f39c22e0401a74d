// fpe_multi.cpp — placeholder (multi-device C ABI lands in a later commit of this round).
