// bin/ckdmip_sw: stand-in for the external CKDMIP shortwave tool as far as the reference's scripts use it (ckdmip.hpp)
#include "ckdmip.hpp"

int main(int argc, char** argv) { return ckdmip_main(argc, argv, true); }
