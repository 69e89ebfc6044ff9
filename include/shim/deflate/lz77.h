/* algorithms/deflate/lz77.h: everything it declares for the hot path lives in mi_deflate.h */
#include "../../mi_deflate.h"
