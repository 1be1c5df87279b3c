// sprspr: dense abundance table on stdin -> sparse table on stdout (sprspr/sprspr.go).
#include "frackyfrac_amd.h"
int main(int argc, char **argv) { return ff_sprspr_main(argc, argv); }
