/* cli_common.h -- shared driver of the six drop-in tools (dexta undexta dexar undexar dexqv undexqv). */
#ifndef CLI_COMMON_H
#define CLI_COMMON_H

enum { TOOL_DEXTA = 0, TOOL_UNDEXTA, TOOL_DEXAR, TOOL_UNDEXAR, TOOL_DEXQV, TOOL_UNDEXQV };

int dex_tool_main(int tool, int argc, char *argv[]);

#endif
