/* algorithms/lz77/lz77.h reduced to one line: lz77/main.c says #include "lz77.h" and compiles unchanged */
#include "../../mi_lz77.h"
