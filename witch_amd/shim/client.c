/* Thin client of the witch-hip GPU server: installed twice, as `hmmsearch` and `hmmalign`
 * (the tool is argv[0]'s basename), so that WITCH's own plug-in keys hmmsearchpath /
 * hmmalignpath (witch_msa/default.config:15-17) can point at them.  Sends
 *     tool \0 cwd \0 arg \0 arg ...
 * over the UNIX socket $WITCH_HIP_SOCKET (default $XDG_RUNTIME_DIR/witch_hip/server.sock or
 * /tmp/witch_hip_<uid>/server.sock, a directory private to the user), prints the
 * reply body and exits with the reply status.  If nothing listens it starts
 * `python3 -m witch_amd.shim.server --daemonize` (repo root = two directories above this binary's
 * directory, or $WITCH_HIP_ROOT) and retries for up to WITCH_HIP_START_TIMEOUT seconds (default 120:
 * the first import of torch on a fresh machine takes a minute).
 * Build: make -C witch_amd/shim */
#include <errno.h>
#include <libgen.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

static int try_connect(const char *path) {
  int fd = socket(AF_UNIX, SOCK_STREAM, 0);
  if (fd < 0) return -1;
  struct sockaddr_un sa;
  memset(&sa, 0, sizeof sa);
  sa.sun_family = AF_UNIX;
  strncpy(sa.sun_path, path, sizeof sa.sun_path - 1);
  if (connect(fd, (struct sockaddr *)&sa, sizeof sa) != 0) { close(fd); return -1; }
  return fd;
}

static void start_server(const char *self, const char *sock) {
  char root[PATH_MAX];
  const char *env = getenv("WITCH_HIP_ROOT");
  if (env) {
    snprintf(root, sizeof root, "%s", env);
  } else {
    char real[PATH_MAX];
    if (!realpath(self, real)) snprintf(real, sizeof real, "%s", self);
    /* <root>/witch_amd/shim/bin/<tool> */
    char *d = dirname(real); d = dirname(d); d = dirname(d); d = dirname(d);
    snprintf(root, sizeof root, "%s", d);
  }
  pid_t pid = fork();
  if (pid == 0) {
    const char *old = getenv("PYTHONPATH");
    char pp[2 * PATH_MAX];
    if (old && *old) snprintf(pp, sizeof pp, "%s:%s", root, old); else snprintf(pp, sizeof pp, "%s", root);
    setenv("PYTHONPATH", pp, 1);
    const char *py = getenv("WITCH_HIP_PYTHON");
    execlp(py ? py : "python3", py ? py : "python3", "-m", "witch_amd.shim.server", "--socket", sock, "--daemonize", (char *)NULL);
    _exit(127);
  }
  if (pid > 0) { int st; waitpid(pid, &st, 0); }
}

int main(int argc, char **argv) {
  char selfbuf[PATH_MAX];
  snprintf(selfbuf, sizeof selfbuf, "%s", argv[0]);
  const char *tool = basename(selfbuf);
  char sockbuf[108];
  const char *sock = getenv("WITCH_HIP_SOCKET");
  if (!sock) {                                     /* same rule as server.py: a directory private to this user */
    const char *rt = getenv("XDG_RUNTIME_DIR");
    if (rt && *rt) snprintf(sockbuf, sizeof sockbuf, "%s/witch_hip/server.sock", rt);
    else snprintf(sockbuf, sizeof sockbuf, "/tmp/witch_hip_%d/server.sock", (int)getuid());
    sock = sockbuf;
  }
  if (argc >= 2 && strcmp(argv[1], "-h") == 0) {
    printf("# %s :: witch-hip level-0 shim (MI355X server behind a UNIX socket)\nUsage: %s [options] <hmmfile> <seqfile>\n", tool, tool);
    return 0;
  }
  int fd = try_connect(sock);
  if (fd < 0) {
    const char *self = argv[0];
    char resolved[PATH_MAX];
    if (!strchr(self, '/')) {                      /* found through PATH */
      ssize_t n = readlink("/proc/self/exe", resolved, sizeof resolved - 1);
      if (n > 0) { resolved[n] = 0; self = resolved; }
    }
    start_server(self, sock);
    const char *te = getenv("WITCH_HIP_START_TIMEOUT");
    int limit = te ? atoi(te) : 120;
    for (int waited = 0; fd < 0 && waited < limit * 10; waited++) {
      struct timespec ts = {0, 100000000};
      nanosleep(&ts, NULL);
      fd = try_connect(sock);
    }
    if (fd < 0) { fprintf(stderr, "%s: cannot reach the witch-hip server at %s (see %s.log)\n", tool, sock, sock); return 1; }
  }
  char cwd[PATH_MAX];
  if (!getcwd(cwd, sizeof cwd)) strcpy(cwd, ".");
  size_t need = strlen(tool) + 1 + strlen(cwd) + 1;
  for (int i = 1; i < argc; i++) need += strlen(argv[i]) + 1;
  char *req = malloc(need), *p = req;
  if (!req) return 1;
  p = stpcpy(p, tool) + 1;
  p = stpcpy(p, cwd) + 1;
  for (int i = 1; i < argc; i++) p = stpcpy(p, argv[i]) + 1;
  size_t off = 0;
  while (off < need) {
    ssize_t n = write(fd, req + off, need - off);
    if (n <= 0) { if (errno == EINTR) continue; perror("write"); return 1; }
    off += (size_t)n;
  }
  shutdown(fd, SHUT_WR);
  char buf[65536];
  int status = 1, have_status = 0;
  char head[32]; size_t hl = 0;
  for (;;) {
    ssize_t n = read(fd, buf, sizeof buf);
    if (n < 0 && errno == EINTR) continue;
    if (n <= 0) break;
    size_t s = 0;
    if (!have_status) {
      while (s < (size_t)n && buf[s] != '\n' && hl + 1 < sizeof head) head[hl++] = buf[s++];
      if (s < (size_t)n && buf[s] == '\n') { head[hl] = 0; status = atoi(head); have_status = 1; s++; }
    }
    if (have_status && s < (size_t)n) fwrite(buf + s, 1, (size_t)n - s, status == 0 ? stdout : stderr);
  }
  close(fd);
  return have_status ? status : 1;
}
