/* algorithms/huffman/huffman.h reduced to one line: huffman/main.c compiles unchanged */
#include "../../mi_huffman.h"
