/*
 * or_sim.c -- ORACLE (test infrastructure): Monte-Carlo frame driver restating
 * src/LDPC_BER_Sim.cpp:121-155,246-311.  Filled in together with the device front end.
 */
#include "or_internal.h"
