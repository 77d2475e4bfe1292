// ABI version of libmicrodit_hip.so (include/microdit_hip.h: MD_ABI_VERSION); the binding refuses a library that disagrees.
#include "../../include/microdit_hip.h"

extern "C" int md_abi_version(void) { return MD_ABI_VERSION; }
