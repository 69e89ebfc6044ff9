/* algorithms/deflate/deflate.h (+ lz77.h) reduced to one line: deflate/main.c compiles unchanged */
#include "../../mi_deflate.h"
