// TEST SCAFFOLDING (oracle/_ref build): the three symbols include/build_version.h declares.
#include "build_version.h"
const char* sortmerna_build_compile_date = "oracle";
const char* sortmerna_build_git_sha = "0";
const char* sortmerna_build_git_date = "0";
