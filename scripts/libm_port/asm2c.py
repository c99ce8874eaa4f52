#!/usr/bin/env python3
"""Transliterates the scalar-double x86-64 code of one libm function (objdump -d --no-show-raw-insn) into C, instruction by
instruction: every xmm register becomes a 64-bit variable (low lane), every fused multiply-add stays ONE fma(), every other
operation one IEEE operation, every jump a goto.  Used for glibc 2.35's __sin_fma / __cos_fma / __ieee754_atan2_fma - the
functions CPython's math.sin / math.cos / math.atan2 resolve to on an FMA-capable x86-64 host - so that the device's 2D steer
(rrt_star_2d.py:67-78) is bit-identical to the reference's.   python asm2c.py <libm.so> <name>:<start>:<stop> [...]  -> C on stdout
"""
import re
import struct
import subprocess
import sys

LIBM = sys.argv[1]
JOBS = [(a.split(":")[0], int(a.split(":")[1], 16), int(a.split(":")[2], 16)) for a in sys.argv[2:]]
data = open(LIBM, "rb").read()
consts = {}      # vaddr -> 8 bytes used as scalar constants
tables = set()   # vaddrs loaded with lea (table bases)
BODIES = []


def disassemble(START, STOP):
  return subprocess.run(["objdump", "-d", "--no-show-raw-insn", "--start-address=%#x" % START, "--stop-address=%#x" % STOP, LIBM],
                     capture_output=True, text=True, check=True).stdout


def vaddr_bytes(addr, n):          # (libm's loadable segments are mapped at file offset == vaddr up to .data; checked below)
    b = data[addr:addr + n]
    return b + b"\0" * (n - len(b))


R64 = ["rax", "rbx", "rcx", "rdx", "rsi", "rdi", "rbp", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "r15"]
R32 = {"eax": "rax", "ebx": "rbx", "ecx": "rcx", "edx": "rdx", "esi": "rsi", "edi": "rdi", "ebp": "rbp"}
R32.update({"r%dd" % i: "r%d" % i for i in range(8, 16)})
R8L = {"al": "rax", "bl": "rbx", "cl": "rcx", "dl": "rdx", "sil": "rsi", "dil": "rdi", "bpl": "rbp"}
R8L.update({"r%db" % i: "r%d" % i for i in range(8, 16)})
R8H = {"ah": "rax", "bh": "rbx", "ch": "rcx", "dh": "rdx"}


def split_ops(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def is_xmm(o):
    return o.startswith("%xmm")


def xr(o):
    return "x%d" % int(o[4:])


def mem_addr(o, target):
    """C expression of the address of a memory operand (vaddr space for rip-relative / table accesses, 'S+off' for the stack)"""
    m = re.match(r"^(-?0x[0-9a-f]+|-?\d+)?\((%\w+)?(?:,(%\w+),(\d))?\)$", o)
    if not m:
        m2 = re.match(r"^%fs:(0x[0-9a-f]+)$", o)
        if m2:
            return ("fs", None)
        raise ValueError("mem operand " + o)
    disp = int(m.group(1), 0) if m.group(1) else 0
    base, idx, sc = m.group(2), m.group(3), m.group(4)
    if base == "%rip":
        return ("abs", target)
    if base == "%rsp":
        assert idx is None
        return ("stk", disp)
    e = "(int64_t)%s" % base[1:] if base else "0"
    if idx:
        e += " + (int64_t)%s * %s" % (idx[1:], sc)
    if disp:
        e += " + %d" % disp
    return ("dyn", e)


def ld64(o, target):
    k, v = mem_addr(o, target)
    if k == "abs":
        consts[v] = vaddr_bytes(v, 8)
        return "K_%x" % v
    if k == "stk":
        return "S64(%d)" % v
    if k == "fs":
        return "0ull"
    return "LD64(%s)" % v


def ireg_read(o, width=None):
    r = o[1:]
    if r in R64:
        return r
    if r in R32:
        return "(uint32_t)%s" % R32[r]
    if r in R8L:
        return "(uint8_t)%s" % R8L[r]
    if r in R8H:
        return "(uint8_t)(%s >> 8)" % R8H[r]
    raise ValueError("reg " + o)


def ireg_write(o, expr):
    r = o[1:]
    if r in R64:
        return "%s = (uint64_t)(%s);" % (r, expr)
    if r in R32:
        return "%s = (uint64_t)(uint32_t)(%s);" % (R32[r], expr)
    if r in R8L:
        return "%s = (%s & ~0xffull) | (uint8_t)(%s);" % (R8L[r], R8L[r], expr)
    if r in R8H:
        return "%s = (%s & ~0xff00ull) | ((uint64_t)(uint8_t)(%s) << 8);" % (R8H[r], R8H[r], expr)
    raise ValueError("reg " + o)


def width_of(o):
    r = o[1:]
    return 64 if r in R64 else 32 if r in R32 else 8


def ival(o, target, width):
    """integer source operand -> C expression (zero-extended to uint64_t)"""
    if o.startswith("$"):
        v = int(o[1:], 0)
        mask = (1 << width) - 1
        return "0x%xull" % (v & mask)
    if o.startswith("%fs:"):
        return "0ull"
    if o.startswith("%"):
        return "(uint64_t)%s" % ireg_read(o)
    k, v = mem_addr(o, target)
    if k == "stk":
        return "S%d(%d)" % (width if width in (32, 64) else 32, v)
    if k == "fs":
        return "0ull"
    if k == "abs":
        consts[v] = vaddr_bytes(v, 8)
        return "(K_%x & 0x%xull)" % (v, (1 << width) - 1)
    return "LD%d(%s)" % (width, v)


out = []


def emit(s):
    out.append(s)


SX = {8: "int8_t", 32: "int32_t", 64: "int64_t"}
UX = {8: "uint8_t", 32: "uint32_t", 64: "uint64_t"}


def setflags_cmp(a, b, w):      # flags of a - b
    emit("fa = (uint64_t)(%s)(%s); fb = (uint64_t)(%s)(%s); fk = 0; fw = %d;" % (UX[w], a, UX[w], b, w))


def setflags_res(r, w):         # logical result: ZF / SF, CF = OF = 0
    emit("fa = (uint64_t)(%s)(%s); fb = 0; fk = 0; fw = %d;" % (UX[w], r, w))


def cond(cc):
    s = "(fw == 64 ? (int64_t)fa : fw == 32 ? (int64_t)(int32_t)fa : (int64_t)(int8_t)fa)"
    t = "(fw == 64 ? (int64_t)fb : fw == 32 ? (int64_t)(int32_t)fb : (int64_t)(int8_t)fb)"
    return {"e": "(fk ? fda == fdb : fa == fb)", "ne": "(fk ? fda != fdb : fa != fb)",
            "g": "(%s > %s)" % (s, t), "ge": "(%s >= %s)" % (s, t), "l": "(%s < %s)" % (s, t), "le": "(%s <= %s)" % (s, t),
            "a": "(fk ? fda > fdb : fa > fb)", "ae": "(fk ? fda >= fdb : fa >= fb)", "b": "(fk ? fda < fdb : fa < fb)",
            "be": "(fk ? fda <= fdb : fa <= fb)", "s": "(%s - %s < 0)" % (s, t), "ns": "(%s - %s >= 0)" % (s, t), "p": "(fk && (fda != fda || fdb != fdb))",
            "np": "(!(fk && (fda != fda || fdb != fdb)))"}[cc]


FMA = {"vfmadd132sd": "fma(D(%(d)s), %(c)s, %(b)s)", "vfmadd213sd": "fma(%(b)s, D(%(d)s), %(c)s)", "vfmadd231sd": "fma(%(b)s, %(c)s, D(%(d)s))",
       "vfnmadd132sd": "fma(-D(%(d)s), %(c)s, %(b)s)", "vfnmadd213sd": "fma(-(%(b)s), D(%(d)s), %(c)s)", "vfnmadd231sd": "fma(-(%(b)s), %(c)s, D(%(d)s))",
       "vfmsub132sd": "fma(D(%(d)s), %(c)s, -(%(b)s))", "vfmsub213sd": "fma(%(b)s, D(%(d)s), -(%(c)s))", "vfmsub231sd": "fma(%(b)s, %(c)s, -D(%(d)s))",
       "vfnmsub132sd": "fma(-D(%(d)s), %(c)s, -(%(b)s))", "vfnmsub213sd": "fma(-(%(b)s), D(%(d)s), -(%(c)s))", "vfnmsub231sd": "fma(-(%(b)s), %(c)s, -D(%(d)s))"}


def dsrc(o, target):      # double-valued source operand
    if is_xmm(o):
        return "D(%s)" % xr(o)
    return "DB(%s)" % ld64(o, target)


def bsrc(o, target):      # 64-bit pattern source operand
    if is_xmm(o):
        return xr(o)
    return ld64(o, target)



def translate(NAME, START, STOP):
    global out
    out = []
    asm = disassemble(START, STOP)
    ins = []
    for line in asm.splitlines():
        m = re.match(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if not m:
            continue
        addr, op, rest = int(m.group(1), 16), m.group(2), m.group(3)
        cm = re.search(r"#\s*([0-9a-f]+)", rest)
        target = int(cm.group(1), 16) if cm else None
        rest = rest.split("#")[0].strip()
        rest = re.sub(r"<[^>]*>", "", rest).strip()
        ins.append((addr, op, rest, target))

    labels = set()
    for addr, op, rest, target in ins:
        if op.startswith("j") and re.match(r"^[0-9a-f]+$", rest.split()[0] if rest else ""):
            labels.add(int(rest.split()[0], 16))


    for addr, op, rest, target in ins:
        if addr in labels:
            emit("L_%x: ;" % addr)
        ops = split_ops(rest)
        try:
            if op in ("nop", "nopl", "nopw", "endbr64", "cs", "push", "pop", "xchg", "data16"):
                continue
            if op in ("sub", "add") and len(ops) == 2 and ops[1] == "%rsp":
                continue
            if op in ("vaddsd", "vsubsd", "vmulsd", "vdivsd"):
                c = {"vaddsd": "+", "vsubsd": "-", "vmulsd": "*", "vdivsd": "/"}[op]
                emit("%s = B(%s %s %s);" % (xr(ops[2]), dsrc(ops[1], target), c, dsrc(ops[0], target)))
            elif op in FMA:
                emit("%s = B(%s);" % (xr(ops[2]), FMA[op] % {"d": xr(ops[2]), "b": dsrc(ops[1], target), "c": dsrc(ops[0], target)}))
            elif op in ("vmovsd", "vmovapd", "vmovq", "vmovaps", "vmovdqa"):
                if len(ops) == 3:
                    emit("%s = %s;" % (xr(ops[2]), xr(ops[0])))
                elif is_xmm(ops[1]):
                    if is_xmm(ops[0]):
                        emit("%s = %s;" % (xr(ops[1]), xr(ops[0])))
                    elif ops[0].startswith("%"):
                        emit("%s = %s;" % (xr(ops[1]), ireg_read(ops[0])))
                    else:
                        emit("%s = %s;" % (xr(ops[1]), ld64(ops[0], target)))
                elif is_xmm(ops[0]):
                    if ops[1].startswith("%"):
                        emit(ireg_write(ops[1], xr(ops[0])))
                    else:
                        k, v = mem_addr(ops[1], target)
                        assert k == "stk", (op, rest)
                        emit("W64(%d, %s);" % (v, xr(ops[0])))
                else:
                    raise ValueError("mov form")
            elif op in ("vandpd", "vorpd", "vxorpd", "vandnpd"):
                a, b = bsrc(ops[0], target), bsrc(ops[1], target)
                e = {"vandpd": "%s & %s" % (b, a), "vorpd": "%s | %s" % (b, a), "vxorpd": "%s ^ %s" % (b, a), "vandnpd": "~%s & %s" % (b, a)}[op]
                emit("%s = %s;" % (xr(ops[2]), e))
            elif op in ("vcomisd", "vucomisd"):
                emit("fda = D(%s); fdb = %s; fk = 1;" % (xr(ops[1]), dsrc(ops[0], target)))
            elif op.startswith("vcmp") and op.endswith("sd"):
                pred = op[4:-2]
                a, b = dsrc(ops[1], target), dsrc(ops[0], target)
                e = {"lt": "%s < %s", "le": "%s <= %s", "nlt": "!(%s < %s)", "nle": "!(%s <= %s)", "eq": "%s == %s", "neq": "%s != %s"}[pred] % (a, b)
                emit("%s = (%s) ? ~0ull : 0ull;" % (xr(ops[2]), e))
            elif op == "vblendvpd":
                emit("%s = (%s >> 63) ? %s : %s;" % (xr(ops[3]), xr(ops[0]), bsrc(ops[1], target), xr(ops[2])))
            elif op == "vcvttsd2si":
                w = width_of(ops[1])
                emit(ireg_write(ops[1], "(%s)%s" % (SX[w], dsrc(ops[0], target))))
            elif op in ("vstmxcsr",):
                k, v = mem_addr(ops[0], target)
                emit("W32(%d, 0x1f80u);" % v)
            elif op in ("vldmxcsr",):
                emit("/* ldmxcsr */;")
            elif op in ("mov", "movl", "movq"):
                src, dst = ops
                if dst.startswith("%"):
                    w = width_of(dst)
                    emit(ireg_write(dst, ival(src, target, w)))
                else:
                    k, v = mem_addr(dst, target)
                    assert k == "stk", (op, rest)
                    w = 32 if op == "movl" or (src.startswith("%") and width_of(src) == 32) else 64
                    emit("W%d(%d, %s);" % (w, v, ival(src, target, w)))
            elif op == "movslq":
                emit(ireg_write(ops[1], "(int64_t)(int32_t)(%s)" % ival(ops[0], target, 32)))
            elif op == "cltq":
                emit("rax = (uint64_t)(int64_t)(int32_t)rax;")
            elif op == "lea":
                k, v = mem_addr(ops[0], target)
                if k == "abs":
                    tables.add(v)
                    emit(ireg_write(ops[1], "0x%xull" % v))
                else:
                    assert k == "dyn", (op, rest)
                    emit(ireg_write(ops[1], v))
            elif op in ("cmp", "cmpl", "cmpq"):
                w = width_of(ops[1]) if ops[1].startswith("%") else (32 if op != "cmpq" else 64)
                setflags_cmp(ival(ops[1], target, w), ival(ops[0], target, w), w)
            elif op in ("test", "testb", "testl"):
                w = width_of(ops[1]) if ops[1].startswith("%") else (8 if op == "testb" else 32)
                setflags_res("(%s) & (%s)" % (ival(ops[1], target, w), ival(ops[0], target, w)), w)
            elif op in ("and", "or", "xor", "add", "sub", "shl", "sar", "shr"):
                dst = ops[1] if len(ops) == 2 else ops[0]
                src = ops[0] if len(ops) == 2 else "$1"
                w = width_of(dst)
                a, b = ival(dst, target, w), ival(src, target, w)
                if op == "sub":
                    emit("{ uint64_t a_ = %s, b_ = %s;" % (a, b))
                    emit(ireg_write(dst, "a_ - b_"))
                    setflags_cmp("a_", "b_", w)
                    emit("}")
                    continue
                e = {"and": "%s & %s", "or": "%s | %s", "xor": "%s ^ %s", "add": "%s + %s", "shl": "%s << %s", "shr": "%s >> %s"}.get(op)
                if op == "sar":
                    e = "(uint64_t)((%s)(%s)(%s) >> %s)" % (SX[w], UX[w], a, b)
                else:
                    e = e % ("(%s)(%s)" % (UX[w], a), b)
                emit(ireg_write(dst, e))
                setflags_res(ireg_read(dst), w)
            elif op == "cmovne":
                emit("if (%s) { %s }" % (cond("ne"), ireg_write(ops[1], ival(ops[0], target, width_of(ops[1])))))
            elif op == "jmp":
                emit("goto L_%x;" % int(ops[0], 16))
            elif op.startswith("j"):
                emit("if (%s) goto L_%x;" % (cond(op[1:]), int(ops[0], 16)))
            elif op == "ret":
                emit("return D(x0);")
            elif op == "call":
                emit("UNSUPPORTED(\"call %s\");" % rest)
            else:
                emit("UNSUPPORTED(\"%s %s\");" % (op, rest))
        except Exception as e:      # something outside the scalar-double subset: only reachable paths matter (checked by the harness)
            emit("UNSUPPORTED(\"%s %s [%s]\");" % (op, rest, e))

    body = ["/* %s: generated by scripts/libm_port/asm2c.py from %s [%#x, %#x) - do not edit */" % (NAME, LIBM.split("/")[-1], START, STOP),
            "LIBM_FN double %s(double arg0, double arg1)\n{" % NAME,
            "    uint64_t %s;" % ", ".join("%s = 0" % r for r in R64),
            "    uint64_t %s;" % ", ".join("x%d = 0" % i for i in range(16)),
            "    uint64_t fa = 0, fb = 0; double fda = 0, fdb = 0; int fk = 0, fw = 32; uint64_t stk[16] = {0};",
            "    x0 = B(arg0); x1 = B(arg1); (void)fa; (void)fb; (void)fda; (void)fdb; (void)fk; (void)fw; (void)stk;"]
    body += ["    " + l for l in out] + ["    return D(x0);\n}"]
    BODIES.append("\n".join(body))


for job in JOBS:
    translate(*job)
print("/* constants of %s referenced by the functions below */" % LIBM.split("/")[-1])
for v in sorted(consts):
    print("LIBM_CONST(K_%x, 0x%016xull)" % (v, struct.unpack("<Q", consts[v])[0]))
print("/* table bases: %s */" % ", ".join("%#x" % t for t in sorted(tables)))
for b_ in BODIES:
    print(b_)
