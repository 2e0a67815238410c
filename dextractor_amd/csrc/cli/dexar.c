/* dexar -- drop-in for the reference's dexar (see cli_common.c); all codec work runs on the GPU. */
#include "cli_common.h"

int main(int argc, char *argv[]) { return dex_tool_main(TOOL_DEXAR, argc, argv); }
