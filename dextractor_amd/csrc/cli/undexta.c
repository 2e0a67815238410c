/* undexta -- drop-in for the reference's undexta (see cli_common.c); all codec work runs on the GPU. */
#include "cli_common.h"

int main(int argc, char *argv[]) { return dex_tool_main(TOOL_UNDEXTA, argc, argv); }
