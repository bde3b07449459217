// cafehip -- command-line front end: runs a CAFE script (or stdin) through the host driver.
//   cafehip [-d device] [script]
// Mirrors main.cpp:25-67 of the reference (REPL / script runner) for the commands in scope.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../../include/cafehost.h"

int main(int argc, char** argv)
{
    int device = 0;
    const char* script = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-d") && i + 1 < argc)
            device = atoi(argv[++i]);
        else
            script = argv[i];
    }
    cafehost_session* s = nullptr;
    if (cafehost_create(&s, device, "stdout") != 0) {
        fprintf(stderr, "%s\n", cafehost_last_error());
        return 2;
    }
    int rc = 0;
    if (script) {
        rc = cafehost_run_script(s, script);
        if (rc < 0) fprintf(stderr, "%s\n", cafehost_last_error());
    } else {
        std::string line;
        while (true) {
            printf("# ");
            fflush(stdout);
            if (!std::getline(std::cin, line)) break;
            rc = cafehost_dispatch(s, line.c_str());
            if (rc < 0) fprintf(stderr, "%s\n", cafehost_last_error());
            if (rc == 1) break;
        }
    }
    cafehost_destroy(s);
    return rc < 0 ? 1 : 0;
}
