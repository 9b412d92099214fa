// ber_sim -- drop-in command line of the reference's prog/ber_sim.cpp:46-160:
//   ber_sim -p <params.ini> [-b <basedir>] [-s <seed>] [-c <custom-name>] [-d <device>]
// The simulation runs on the MI355X through liblut_ldpc_amd.so.
#include "ber_sim_driver.hpp"

int main(int argc, char **argv) { return lut_ldpc::ber_sim_main(argc, argv); }
