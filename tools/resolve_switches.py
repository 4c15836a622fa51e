#!/usr/bin/env python3
"""A small `unifdef`: resolves chosen preprocessor switches of the kernel sources to fixed values -- the closed A/B variants
(r06 pruning).  For every NAME=value given:
  * an `#ifndef NAME / #define NAME ... / #endif` default block is deleted,
  * `#if` / `#elif` conditions that mention only the given names are evaluated: the taken branch stays, the others go,
  * remaining uses of NAME in the code become the literal value.
Conditions that mention other identifiers are left alone.      python3 tools/resolve_switches.py FILE... -- NAME=value ..."""
import re
import sys


def resolve(text, vals):
    lines = text.split("\n")
    out = []
    # stack entries: [kind, taken_already, emitting, known]   kind: 'known' (we resolve it) or 'other'
    stack = []

    def emitting():
        return all(e[2] for e in stack)

    def evaluate(expr):
        e = re.sub(r"//.*$", "", expr).strip()
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in vals else "defined(%s)" % m.group(1), e)
        names = set(re.findall(r"[A-Za-z_]\w*", e))
        if not names or not names <= set(vals):
            return None
        for n in sorted(names, key=len, reverse=True):
            e = re.sub(r"\b%s\b" % n, str(vals[n]), e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        return bool(eval(e))

    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)$", ln)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("ifndef", "ifdef", "if"):
            if d == "if":
                v = evaluate(rest)
            else:
                name = rest.strip().split()[0] if rest.strip() else ""
                # `#ifndef NAME` default block of a resolved name: drop the block whole
                v = None
                if name in vals:
                    v = (d == "ifdef")
            if v is None:
                stack.append(["other", False, True])
                if emitting():
                    out.append(ln)
            else:
                stack.append(["known", v, v])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "other":
                if emitting():
                    out.append(ln)
            else:
                if top[1]:
                    top[2] = False
                else:
                    v = evaluate(rest)
                    if v is None:
                        raise SystemExit(f"cannot resolve #elif at line {i + 1}: {ln}")
                    top[1] = top[2] = v
        elif d == "else":
            top = stack[-1]
            if top[0] == "other":
                if emitting():
                    out.append(ln)
            else:
                top[2] = not top[1]
                top[1] = True
        else:  # endif
            top = stack.pop()
            if top[0] == "other" and emitting():
                out.append(ln)
        i += 1
    res = []
    for ln in out:   # the literal value in the code, not in the comments
        code, sep, comment = ln.partition("//")
        for n, v in vals.items():
            code = re.sub(r"\b%s\b" % n, str(v), code)
        res.append(code + sep + comment)
    return "\n".join(res)


def main():
    args = sys.argv[1:]
    k = args.index("--")
    files, pairs = args[:k], args[k + 1:]
    vals = {}
    for p in pairs:
        n, v = p.split("=")
        vals[n] = int(v)
    for f in files:
        src = open(f).read()
        dst = resolve(src, vals)
        if dst != src:
            open(f, "w").write(dst)
            print(f"{f}: {src.count(chr(10)) - dst.count(chr(10))} lines removed")


if __name__ == "__main__":
    main()
