/* refrun -- sandboxed executor for the reference author's own liblqr build.
 *
 * BUILD CONTAINER ONLY.  Nothing here is product or test code and nothing here travels to the
 * GPU box; what it produces are DATA files (tests/golden/ref_*.npz).
 *
 * /root/reference/windows_installer_files/lqr-pack4win/.zip holds gimp-lqr-plugin.exe, a PE32/i386
 * build of the plug-in statically linked with liblqr (winpack.sh:8,52-57).  This program is a
 * freestanding i386 Linux process (no libc: gcc -m32 -nostdlib -static) that
 *   1. reserves the PE's image range and a heap arena,
 *   2. enters seccomp STRICT mode (only read/write/_exit remain possible),
 *   3. serves a small binary protocol on stdin/stdout: write/read memory, allocate, call a
 *      cdecl function at an address with given stack words, read the progress-event log.
 * The driver (ref_engine.py) copies the PE's sections in, points the import slots the ENGINE
 * code uses (g_try_malloc, g_try_malloc0, g_free, g_strlcpy, two atomics, g_usleep, pow, _assert)
 * at the functions below and every other slot at a trap, and then calls lqr_* entry points.
 * The exe's own entry point, GIMP/GTK code and CRT are never executed.
 */
typedef unsigned int u32;
typedef unsigned char u8;
typedef unsigned long long u64;

#define SYS_exit 1
#define SYS_read 3
#define SYS_write 4
#define SYS_prctl 172
#define SYS_mmap2 192

static inline int sys3(int n, u32 a, u32 b, u32 c)
{
    int r;
    __asm__ volatile("int $0x80" : "=a"(r) : "0"(n), "b"(a), "c"(b), "d"(c) : "memory");
    return r;
}
static u32 sys_mmap2(u32 addr, u32 len, u32 prot, u32 flags, int fd, u32 pgoff)
{
    u32 r;
    __asm__ volatile("push %%ebp; mov %7, %%ebp; int $0x80; pop %%ebp"
                     : "=a"(r) : "0"(SYS_mmap2), "b"(addr), "c"(len), "d"(prot), "S"(flags), "D"(fd), "g"(pgoff) : "memory");
    return r;
}
static void die(int code) { for (;;) sys3(SYS_exit, code, 0, 0); }

static void rd(void *p, u32 n)
{
    u8 *c = p;
    while (n) {
        int r = sys3(SYS_read, 0, (u32) c, n);
        if (r <= 0) die(r == 0 ? 0 : 90);
        c += r; n -= r;
    }
}
static void wr(const void *p, u32 n)
{
    const u8 *c = p;
    while (n) {
        int r = sys3(SYS_write, 1, (u32) c, n);
        if (r <= 0) die(91);
        c += r; n -= r;
    }
}

/* ---------------- heap: first fit, address-ordered implicit list, canaries ---------------- */
#define ARENA_BASE 0x10000000u
#define ARENA_SIZE 0xA0000000u      /* 2.5 GiB of address space, touched lazily */
#define CANARY 32
typedef struct { u32 size; u32 used; u32 req; u32 magic; } Hdr;      /* size = whole block incl. header */
static u8 *arena, *arena_top, *arena_end;
static u32 n_alloc, n_free, peak, overruns, overrun_req, overrun_off;
static int poison = -1;

static void *heap_alloc(u32 n, int zero)
{
    u32 need = (sizeof(Hdr) + n + CANARY + 15) & ~15u;
    u8 *p = arena;
    if (need < n) return 0;
    while (p < arena_top) {
        Hdr *h = (Hdr *) p;
        if (!h->used) {
            /* coalesce with following free blocks */
            while (p + h->size < arena_top && !((Hdr *) (p + h->size))->used) h->size += ((Hdr *) (p + h->size))->size;
            if (p + h->size == arena_top) { arena_top = p; break; }       /* trailing free block: give it back */
            if (h->size >= need) {
                if (h->size - need >= 64) {
                    Hdr *r = (Hdr *) (p + need);
                    r->size = h->size - need; r->used = 0; r->magic = 0x48454150;
                    h->size = need;
                }
                goto found;
            }
        }
        p += h->size;
    }
    if ((u32) (arena_end - arena_top) < need) return 0;
    p = arena_top; arena_top += need;
    ((Hdr *) p)->size = need;
    ((Hdr *) p)->magic = 0x48454150;
found: {
        Hdr *h = (Hdr *) p;
        u8 *u = p + sizeof(Hdr);
        u32 i;
        h->used = 1; h->req = n;
        if (zero) for (i = 0; i < n; i++) u[i] = 0;
        else if (poison >= 0) for (i = 0; i < n; i++) u[i] = (u8) poison;
        for (i = n; i < h->size - sizeof(Hdr); i++) u[i] = 0xA5;
        n_alloc++;
        if ((u32) (arena_top - arena) > peak) peak = arena_top - arena;
        return u;
    }
}
static void heap_free(void *u)
{
    Hdr *h;
    if (!u) return;
    h = (Hdr *) ((u8 *) u - sizeof(Hdr));
    if (h->magic != 0x48454150 || !h->used) die(92);      /* bad or double free */
    {   /* did the engine write past the end of this block while it owned it? */
        u8 *c = (u8 *) u; u32 i;
        for (i = h->req; i < h->size - sizeof(Hdr); i++)
            if (c[i] != 0xA5) { if (!overruns++) { overrun_req = h->req; overrun_off = i; } break; }
    }
    h->used = 0;
    n_free++;
}
/* returns number of blocks whose canary was overwritten; *first = user address of the first */
static u32 heap_check(u32 *first, u32 *first_req, u32 *first_off)
{
    u8 *p = arena; u32 bad = 0;
    *first = 0; *first_req = 0; *first_off = 0;
    while (p < arena_top) {
        Hdr *h = (Hdr *) p;
        if (h->magic != 0x48454150 || h->size < sizeof(Hdr)) { if (!bad++) *first = (u32) p; break; }
        if (h->used) {
            u8 *u = p + sizeof(Hdr); u32 i;
            for (i = h->req; i < h->size - sizeof(Hdr); i++)
                if (u[i] != 0xA5) { if (!bad++) { *first = (u32) u; *first_req = h->req; *first_off = i; } break; }
        }
        p += h->size;
    }
    return bad;
}

/* ---------------- what the engine code imports ---------------- */
static void *g_try_malloc(u32 n) { return n ? heap_alloc(n, 0) : 0; }
static void *g_try_malloc0(u32 n) { return n ? heap_alloc(n, 1) : 0; }
static void g_free(void *p) { heap_free(p); }
static u32 g_strlcpy(char *d, const char *s, u32 size)
{
    u32 n = 0;
    while (s[n]) n++;
    if (size) {
        u32 k = n < size - 1 ? n : size - 1, i;
        for (i = 0; i < k; i++) d[i] = s[i];
        d[k] = 0;
    }
    return n;
}
static void g_atomic_int_add(volatile int *a, int v) { *a += v; }
static int g_atomic_int_exchange_and_add(volatile int *a, int v) { int o = *a; *a += v; return o; }
static void g_usleep(u32 us) { (void) us; }
static int pow_calls;
/* msvcrt pow is not available; the engine calls it once, in lqr_carver_init, for |dx|^1.5.
   x^1.5 = x*sqrt(x) in extended precision, rounded once to double: ref_engine.py checks that this is the
   correctly rounded power for every integer 0..64 (delta_x is at most that in anything we generate);
   any other exponent is refused. */
static double c_pow(double x, double y)
{
    volatile double r;
    unsigned short cw_old, cw64 = 0x37f;
    pow_calls++;
    if (y != 1.5 || x < 0) die(93);
    __asm__ volatile("fnstcw %0" : "=m"(cw_old));
    __asm__ volatile("fldcw %0" : : "m"(cw64));
    {
        long double s;       /* 64-bit mantissa, one rounding to double at the store */
        __asm__("fsqrt" : "=t"(s) : "0"((long double) x));
        r = (double) (s * (long double) x);
    }
    __asm__ volatile("fldcw %0" : : "m"(cw_old));
    return r;
}
static void c_assert(const char *msg, const char *file, int line) { (void) msg; (void) file; (void) line; die(94); }
static void trap_import(void) { die(95); }

/* ---------------- progress recorder (LqrProgress callbacks) ---------------- */
#define EV_MAX 65536
typedef struct { u32 kind; double val; char msg[52]; } Ev;
static Ev *evlog; static u32 n_ev;
static void ev_push(u32 kind, double v, const char *m)
{
    Ev *e; u32 i = 0;
    if (n_ev >= EV_MAX) return;
    e = &evlog[n_ev++];
    e->kind = kind; e->val = v;
    if (m) for (; i < 51 && m[i]; i++) e->msg[i] = m[i];
    e->msg[i] = 0;
}
static int prog_ret = 1;       /* LQR_OK */
static int cb_init(const char *m) { ev_push(1, 0, m); return prog_ret; }
static int cb_update(double p) { ev_push(2, p, 0); return prog_ret; }
static int cb_end(const char *m) { ev_push(3, 0, m); return prog_ret; }

/* ---------------- cdecl call with an explicit stack image ---------------- */
unsigned short fpu_cw = 0x37f;     /* what the exe's own _fpreset (CRT_fp10: fninit) leaves: 64-bit mantissa */
/* u32 tramp(u32 fn, const u32 *words, u32 n): copies n words to a 16-byte aligned stack top, resets the
   x87, loads fpu_cw, calls fn.  Declared twice below: integer result in eax, floating result in st(0). */
__asm__(
    ".text\n"
    ".globl tramp_i\n.globl tramp_f\n"
    "tramp_i:\ntramp_f:\n"
    "  push %ebp\n  mov %esp, %ebp\n  push %esi\n  push %edi\n  push %ebx\n"
    "  mov 16(%ebp), %ecx\n"
    "  mov 12(%ebp), %esi\n"
    "  lea (,%ecx,4), %eax\n"
    "  sub %eax, %esp\n"
    "  and $-16, %esp\n"
    "  mov %esp, %edi\n"
    "  cld\n  rep movsl\n"
    "  fninit\n"
    "  fldcw fpu_cw\n"
    "  call *8(%ebp)\n"
    "  lea -12(%ebp), %esp\n"
    "  pop %ebx\n  pop %edi\n  pop %esi\n  pop %ebp\n  ret\n");
extern u32 tramp_i(u32 fn, const u32 *words, u32 n);
extern double tramp_f(u32 fn, const u32 *words, u32 n);

/* ---------------- control-word wrappers ----------------
 * The exe is x87 code: a float expression is evaluated at the precision the control word selects and rounded to float
 * only when stored.  To make the genuine code evaluate its float-ONLY functions (the DP: build_mmap / update_mmap; init's
 * rigidity table; transpose's rescaling of it; inflate's averages) the way an SSE2 build of the same source does (every
 * operation rounded to float), the driver retargets the call sites of such a function to one of these stubs, which
 * runs the original under another control word (0x07f: 24-bit mantissa) and restores the caller's.  Not re-entrant per
 * slot, which the wrapped functions do not need. */
#define N_WRAP 8
u32 wrap_orig[N_WRAP], wrap_ret[N_WRAP];
unsigned short wrap_cw[N_WRAP], wrap_saved[N_WRAP];
#define WRAP_STUB(k) \
    __asm__(".text\n.globl wrap_stub" #k "\nwrap_stub" #k ":\n" \
            "  popl wrap_ret+4*" #k "\n" \
            "  fnstcw wrap_saved+2*" #k "\n" \
            "  fldcw wrap_cw+2*" #k "\n" \
            "  call *wrap_orig+4*" #k "\n" \
            "  fldcw wrap_saved+2*" #k "\n" \
            "  jmp *wrap_ret+4*" #k "\n"); \
    extern void wrap_stub##k(void);
WRAP_STUB(0) WRAP_STUB(1) WRAP_STUB(2) WRAP_STUB(3) WRAP_STUB(4) WRAP_STUB(5) WRAP_STUB(6) WRAP_STUB(7)

/* ---------------- protocol ---------------- */
enum { OP_WRITE = 1, OP_READ, OP_ALLOC, OP_FREE, OP_CALL, OP_INFO, OP_SETCW, OP_EVENTS, OP_HEAPCHECK, OP_SCANALL, OP_POISON,
       OP_PROGRET, OP_QUIT, OP_WRAP };
#define IMAGE_BASE 0x400000u
#define IMAGE_SIZE 0x100000u

static u32 words[64];

void _start_c(void)
{
    u32 hdr[4];
    if (sys_mmap2(IMAGE_BASE, IMAGE_SIZE, 7, 0x32 /* PRIVATE|FIXED|ANON */, -1, 0) != IMAGE_BASE) die(80);
    arena = (u8 *) sys_mmap2(ARENA_BASE, ARENA_SIZE, 3, 0x4032 /* + NORESERVE */, -1, 0);
    if ((u32) arena != ARENA_BASE) die(81);
    arena_top = arena; arena_end = arena + ARENA_SIZE;
    evlog = (Ev *) sys_mmap2(0, EV_MAX * sizeof(Ev), 3, 0x22, -1, 0);
    if ((u32) evlog > 0xfffff000u) die(82);
    if (sys3(SYS_prctl, 22 /* PR_SET_SECCOMP */, 1 /* SECCOMP_MODE_STRICT */, 0) != 0) die(83);
    for (;;) {
        rd(hdr, 16);
        switch (hdr[0]) {
        case OP_WRITE: rd((void *) hdr[1], hdr[2]); { u32 ok = 1; wr(&ok, 4); } break;
        case OP_READ: wr((void *) hdr[1], hdr[2]); break;
        case OP_ALLOC: { u32 p = (u32) heap_alloc(hdr[1], hdr[2]); wr(&p, 4); } break;
        case OP_FREE: { u32 ok = 1; heap_free((void *) hdr[1]); wr(&ok, 4); } break;
        case OP_CALL: {        /* hdr[1] fn, hdr[2] n words, hdr[3] 0 = int result, 1 = floating */
            struct { u32 eax; double st0; } __attribute__((packed)) res;
            if (hdr[2] > 64) die(84);
            rd(words, hdr[2] * 4);
            res.eax = 0; res.st0 = 0;
            if (hdr[3]) res.st0 = tramp_f(hdr[1], words, hdr[2]);
            else res.eax = tramp_i(hdr[1], words, hdr[2]);
            wr(&res, 12);
        } break;
        case OP_INFO: {
            u32 t[16] = { (u32) g_try_malloc, (u32) g_try_malloc0, (u32) g_free, (u32) g_strlcpy, (u32) g_atomic_int_add,
                          (u32) g_atomic_int_exchange_and_add, (u32) g_usleep, (u32) c_pow, (u32) c_assert,
                          (u32) trap_import, (u32) cb_init, (u32) cb_update, (u32) cb_end, n_alloc, n_free, peak };
            wr(t, sizeof t);
        } break;
        case OP_SETCW: { u32 ok = 1; fpu_cw = (unsigned short) hdr[1]; wr(&ok, 4); } break;
        case OP_EVENTS: { u32 n = n_ev; wr(&n, 4); wr(evlog, n * sizeof(Ev)); n_ev = 0; } break;
        case OP_HEAPCHECK: {
            u32 r[8]; r[0] = heap_check(&r[1], &r[2], &r[3]);
            r[4] = overruns; r[5] = overrun_req; r[6] = overrun_off; r[7] = 0; wr(r, 32);
        } break;
        case OP_SCANALL: {
            /* the read-out loop of io_functions.c:155-164 run in here for speed:
               hdr[1] = &lqr_carver_scan_line, hdr[2] = carver, hdr[3] = bytes per line;
               reply: n lines, then per line {index, bytes} */
            u32 scratch[2], cnt = 0, a[3];
            a[0] = hdr[2]; a[1] = (u32) &scratch[0]; a[2] = (u32) &scratch[1];
            for (;;) {
                u32 more = tramp_i(hdr[1], a, 3);
                wr(&more, 4);
                if (!more) break;
                wr(&scratch[0], 4);
                wr((void *) scratch[1], hdr[3]);
                cnt++;
            }
        } break;
        case OP_POISON: { u32 ok = 1; poison = (int) hdr[1]; wr(&ok, 4); } break;
        case OP_PROGRET: { u32 ok = 1; prog_ret = (int) hdr[1]; wr(&ok, 4); } break;
        case OP_WRAP: {       /* hdr[1] slot, hdr[2] original function, hdr[3] control word; reply: the stub's address */
            static void (*const stubs[N_WRAP])(void) = { wrap_stub0, wrap_stub1, wrap_stub2, wrap_stub3, wrap_stub4, wrap_stub5,
                                                         wrap_stub6, wrap_stub7 };
            u32 a;
            if (hdr[1] >= N_WRAP) die(86);
            wrap_orig[hdr[1]] = hdr[2]; wrap_cw[hdr[1]] = (unsigned short) hdr[3];
            a = (u32) stubs[hdr[1]];
            wr(&a, 4);
        } break;
        case OP_QUIT: die(0);
        default: die(85);
        }
    }
}
__asm__(".text\n.globl _start\n_start:\n  xor %ebp, %ebp\n  and $-16, %esp\n  call _start_c\n  hlt\n");
