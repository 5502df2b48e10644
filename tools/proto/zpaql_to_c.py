#!/usr/bin/env python3
"""Design check for DESIGN.md section 7-2(a): translate a ZPAQL program (HCOMP or PCOMP bytecode) into straight-line C
once, instead of interpreting it byte by byte.  libzpaq does the same for x86 (its JIT, ZSFX/libzpaq.cpp:2709-4261);
on the GPU the generated text would go through hiprtc together with the coder kernel.  Every jump target of ZPAQL is
static (JT/JF/JMP are relative to the instruction, LJ is absolute), so the translation is one label per instruction
and plain gotos.  Semantics follow ZPAQL::run0/execute (ZSFX/libzpaq.cpp:1033-1254): 32-bit wrap-around, division and
modulo by zero give 0, shifts use the count mod 32, *B<>A swaps only the low byte of A, M/H indices are masked.

    emit_c(code, name) -> C source of
        void NAME(uint32_t input, struct zvm* z)      // z: a b c d f, r[256], h[], hmask, m[], mmask, out callback

tests/test_zpaql_to_c_cpu.py compiles the result with gcc and runs it against the reference VM."""
import sys

REGS = ["z->a", "z->b", "z->c", "z->d"]


def operand(sel, code, pc):
    """value expression of operand sel (0..7) -> (expr, new pc)"""
    if sel < 4:
        return REGS[sel], pc
    if sel == 4:
        return "z->m[z->b & z->mmask]", pc
    if sel == 5:
        return "z->m[z->c & z->mmask]", pc
    if sel == 6:
        return "z->h[z->d & z->hmask]", pc
    return str(code[pc]), pc + 1


def emit_c(code, name="zpaql_run"):
    out = ["#include <stdint.h>",
           "struct zvm { uint32_t a, b, c, d, f, r[256]; uint32_t* h; uint32_t hmask; uint8_t* m; uint32_t mmask;",
           "             void (*out)(void*, int); void* out_arg; int err; };",
           "void %s(uint32_t input, struct zvm* z) {" % name,
           "  uint32_t t; z->a = input;"]
    n = len(code)
    # instruction boundaries reachable linearly from 0 (ZPAQL programs are written that way; a jump into the middle of
    # an instruction gets an error label)
    starts, pc = [], 0
    while pc < n:
        starts.append(pc)
        op = code[pc]
        pc += 3 if op == 255 else 2 if (op & 7) == 7 else 1
    valid = set(starts)

    def jump(target):
        return "goto L%d;" % target if target in valid else "{ z->err = 1; return; }"

    for pc in starts:
        op = code[pc]
        nxt = pc + (3 if op == 255 else 2 if (op & 7) == 7 else 1)
        s = None
        if op == 56:
            s = "return;"
        elif op == 0:
            s = "{ z->err = 1; return; }"
        elif op == 255:
            s = jump(code[pc + 1] + 256 * code[pc + 2]) if pc + 2 < n else "{ z->err = 1; return; }"
        elif op in (39, 47, 63):
            off = ((code[pc + 1] + 128) & 255) - 128
            tgt = nxt + off
            cond = {39: "if (z->f) ", 47: "if (!z->f) ", 63: ""}[op]
            s = cond + jump(tgt)
        elif op == 55:
            s = "z->r[%d] = z->a;" % code[pc + 1]
        elif op == 57:
            s = "if (z->out) z->out(z->out_arg, (int)(z->a & 255));"
        elif op == 59:
            s = "z->a = (z->a + z->m[z->b & z->mmask] + 512) * 773;"
        elif op == 60:
            s = "z->h[z->d & z->hmask] = (z->h[z->d & z->hmask] + z->a + 512) * 773;"
        elif op < 56:
            g, k = op >> 3, op & 7
            tgt = [REGS[0], REGS[1], REGS[2], REGS[3], "z->m[z->b & z->mmask]", "z->m[z->c & z->mmask]", "z->h[z->d & z->hmask]"][g] if g < 7 else None
            if tgt is not None:
                if k == 0 and g:                       # X<>A
                    if g in (4, 5):                    # a byte of M swaps with the low byte of A only
                        s = "t = %s; %s = (uint8_t)z->a; z->a = (z->a & 0xffffff00u) | t;" % (tgt, tgt)
                    else:
                        s = "t = %s; %s = z->a; z->a = t;" % (tgt, tgt)
                elif k == 1:
                    s = "++%s;" % tgt
                elif k == 2:
                    s = "--%s;" % tgt
                elif k == 3:
                    s = "%s = ~%s;" % (tgt, tgt)
                elif k == 4:
                    s = "%s = 0;" % tgt
                elif k == 7 and g < 4:
                    s = "%s = z->r[%d];" % (tgt, code[pc + 1])
        elif 64 <= op < 120:
            g, sel = (op - 64) >> 3, op & 7
            v, _ = operand(sel, code, pc + 1)
            tgt = [REGS[0], REGS[1], REGS[2], REGS[3], "z->m[z->b & z->mmask]", "z->m[z->c & z->mmask]", "z->h[z->d & z->hmask]"][g]
            s = "%s = %s;" % (tgt, v)
        elif 128 <= op < 240:
            k, sel = (op - 128) >> 3, op & 7
            v, _ = operand(sel, code, pc + 1)
            v = "((uint32_t)(%s))" % v
            s = ["z->a += %s;", "z->a -= %s;", "z->a *= %s;", "t = %s; z->a = t ? z->a / t : 0;", "t = %s; z->a = t ? z->a %% t : 0;",
                 "z->a &= %s;", "z->a &= ~%s;", "z->a |= %s;", "z->a ^= %s;", "z->a <<= (%s & 31);", "z->a >>= (%s & 31);",
                 "z->f = z->a == %s;", "z->f = z->a < %s;", "z->f = z->a > %s;"][k] % v
        if s is None:
            s = "{ z->err = 1; return; }"
        out.append("L%d: %s" % (pc, s))
    out.append("  z->err = 1;   /* ran off the end of the program */")
    out.append("}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    sys.stdout.write(emit_c(bytes.fromhex(sys.argv[1])))
