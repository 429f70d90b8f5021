#!/usr/bin/env python3
"""Resolve preprocessor conditionals over macros with known values and drop the dead branches (a small `unifdef`).
usage: unifdef.py -DNAME=VALUE ... file...   (files are rewritten in place)
Only conditionals whose expression consists of known macros, integers and C operators are resolved; every other line is kept
as it is.  Used to retire experiment switches (TR_QNODES, TR_BVH8, ...) once their records are in profiles/ and experiments/."""
import re
import sys


def evaluate(expr, known):
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", lambda m: ("1" if (m.group(1) or m.group(2)) in known else "?"), expr)
    ids = set(re.findall(r"[A-Za-z_]\w*", e))
    if "?" in e or not ids <= set(known):
        return None
    for k in ids:
        e = re.sub(r"\b%s\b" % k, str(known[k]), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(e, {"__builtins__": {}}))


def process(text, known):
    out, stack = [], []      # stack entries: [resolved, taken_already, currently_emitting, parent_emitting]
    emitting = True
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting:
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2).split("//")[0].strip()
        if d in ("if", "ifdef", "ifndef"):
            if d == "if":
                v = evaluate(rest, known)
            else:
                name = rest.split()[0]
                v = (name in known) if name in known else None
                if v is not None and d == "ifndef":
                    v = not v
            parent = emitting
            if v is None:
                stack.append([False, False, emitting, parent])
                if emitting:
                    out.append(line)
            else:
                stack.append([True, v, parent and v, parent])
                emitting = parent and v
        elif d == "elif":
            top = stack[-1]
            if not top[0]:
                if emitting:
                    out.append(line)
                continue
            if top[1]:
                emitting = False
            else:
                v = evaluate(rest, known)
                if v is None:      # becomes the opening #if of what is left
                    top[0] = False
                    emitting = top[3]
                    if emitting:
                        out.append(re.sub(r"#\s*elif", "#if", line, 1))
                else:
                    top[1] = v
                    emitting = top[3] and v
        elif d == "else":
            top = stack[-1]
            if not top[0]:
                if emitting:
                    out.append(line)
                continue
            emitting = top[3] and not top[1]
            top[1] = True
        else:
            top = stack.pop()
            if not top[0] and top[3]:
                out.append(line)
            emitting = top[3]
    return "\n".join(out)


if __name__ == "__main__":
    known, files = {}, []
    for a in sys.argv[1:]:
        if a.startswith("-D"):
            k, _, v = a[2:].partition("=")
            known[k] = int(v or "1")
        else:
            files.append(a)
    for f in files:
        src = open(f).read()
        dst = process(src, known)
        if dst != src:
            open(f, "w").write(dst)
            print(f"{f}: {src.count(chr(10)) - dst.count(chr(10))} lines removed")
