// frcfrc_main.cpp -- entry point of the `frcfrc` executable (frcfrc/frcfrc.go:29).
#include "frackyfrac_amd.h"

int main(int argc, char **argv) { return ff_frcfrc_main(argc, argv); }
