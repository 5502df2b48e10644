/* Stand-in for <windows.h> so the reference's ZSFX/libzpaq.cpp (which includes it
   unconditionally, ZSFX/libzpaq.cpp:28) compiles on Linux with -Dunix.
   The unix branch of allocx() (ZSFX/libzpaq.cpp:57-89) needs mmap. */
#include <sys/mman.h>
