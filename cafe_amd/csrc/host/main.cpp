// cafehip -- command-line front end: runs a CAFE script (or stdin) through the host driver.
//   cafehip [-d device] [script]                 one GPU
//   cafehip --gpus N script                      N processes, one per GPU (devices 0..N-1), families sharded,
//                                                one RCCL all-gather per objective evaluation (include/cafehost.h)
//   cafehip --comm script                        one rank THROUGH the communicator path (functional check)
// Mirrors main.cpp:25-67 of the reference (REPL / script runner) for the commands in scope.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <algorithm>
#include <csignal>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "../../../include/cafehost.h"

static int run(cafehost_session* s, const char* script)
{
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
    return rc;
}

int main(int argc, char** argv)
{
    int device = 0, gpus = 1, rank = -1, world = 0;
    bool comm_one = false, same_device = false;
    const char* script = nullptr;
    const char* id_file = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-d") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
        else if (!strcmp(argv[i], "--comm")) comm_one = true;
        else if (!strcmp(argv[i], "--same-device")) same_device = true;   // every rank on device 0 (functional check on a 1-GPU box)
        else script = argv[i];
    }

    if (rank < 0 && (gpus > 1 || comm_one)) {
        // launcher: create the communicator id, start one child per GPU, wait for all of them
        if (!script) {
            fprintf(stderr, "cafehip: --gpus needs a script file (every rank runs the same commands)\n");
            return 2;
        }
        unsigned char id[CAFEHOST_COMM_ID_BYTES];
        if (cafehost_comm_unique_id(id) != 0) {
            fprintf(stderr, "%s\n", cafehost_last_error());
            return 2;
        }
        char path[] = "/tmp/cafehip_comm_XXXXXX";
        const int fd = mkstemp(path);
        if (fd < 0 || write(fd, id, sizeof id) != (ssize_t)sizeof id) {
            perror("cafehip: communicator id file");
            return 2;
        }
        close(fd);
        std::vector<pid_t> kids;
        for (int r = 0; r < gpus; ++r) {
            const pid_t pid = fork();
            if (pid == 0) {
                const std::string rs = std::to_string(r), ws = std::to_string(gpus), ds = std::to_string(same_device ? device : r);
                execl("/proc/self/exe", argv[0], "-d", ds.c_str(), "--rank", rs.c_str(), "--world", ws.c_str(), "--id-file", path,
                      script, (char*)nullptr);
                perror("cafehip: exec");
                _exit(127);
            }
            kids.push_back(pid);
        }
        // wait for whichever child ends next; one that dies (e.g. before it joins the communicator) would leave the
        // others waiting in ncclCommInitRank or a collective forever: end them too
        // Only OUR children are waited for (one waitpid per live pid, polled: another child of this process is left
        // alone), and a reaped child leaves the list at once -- its pid may be reused by an unrelated process, which must
        // never be signalled.
        int bad = 0;
        std::vector<pid_t> live = kids;
        while (!live.empty()) {
            bool progressed = false;
            for (size_t i = 0; i < live.size();) {
                int st = 0;
                const pid_t k = waitpid(live[i], &st, WNOHANG);
                if (k == 0) {
                    ++i;
                    continue;
                }
                progressed = true;
                live.erase(live.begin() + i);   // reaped (or not ours any more: k < 0)
                if (k < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                    if (!bad)
                        for (pid_t other : live) kill(other, SIGTERM);   // exact pids of children that are still running
                    ++bad;
                }
            }
            if (!progressed) usleep(2000);
        }
        unlink(path);
        if (bad) cafehost_comm_cleanup(id);   // killed ranks cannot remove their shared-memory names: do it for them
        return bad ? 1 : 0;
    }

    cafehost_session* s = nullptr;
    // sharded: every rank executes the script; rank 0 keeps the log
    if (cafehost_create(&s, device, (rank > 0) ? "/dev/null" : "stdout") != 0) {
        fprintf(stderr, "%s\n", cafehost_last_error());
        return 2;
    }
    if (rank >= 0) {
        unsigned char id[CAFEHOST_COMM_ID_BYTES];
        FILE* f = id_file ? fopen(id_file, "rb") : nullptr;
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) {
            fprintf(stderr, "cafehip: cannot read the communicator id from %s\n", id_file ? id_file : "(no --id-file)");
            return 2;
        }
        fclose(f);
        if (cafehost_init_comm(s, rank, world, id) != 0) {
            fprintf(stderr, "%s\n", cafehost_last_error());
            return 2;
        }
    }
    const int rc = run(s, script);
    if (rank == 0) {
        double sec = 0;
        long calls = 0;
        cafehost_exchange_stats(s, &sec, &calls);
        if (calls) fprintf(stderr, "cafehip: %d ranks, %ld exchanges, %.1f us each\n", world, calls, 1e6 * sec / calls);
    }
    cafehost_destroy(s);
    return rc < 0 ? 1 : 0;
}
