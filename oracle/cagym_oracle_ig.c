/* cagym_oracle_ig.c -- information-gain primitives of the parity oracle (filled in below). */
#include "cagym_oracle.h"
