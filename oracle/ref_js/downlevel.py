#!/usr/bin/env python3
"""TypeScript -> plain ECMAScript 2019 modules, for running the REFERENCE's own source on the Node 12 of this image.

TEST INFRASTRUCTURE (oracle/): the product never imports anything produced here.  The reference (paulmillr/noble-curves,
/root/reference/src) is TypeScript that Node >= 20.19 executes directly; this image has Node 12.22 and no TypeScript
compiler.  This script reads the reference's files WHERE THEY LIE and writes type-stripped copies into a directory that
oracle/refjs.py packs into oracle/_ref/refjs.bundle (a gzip tar: git-ignored, never committed - reference sources do not enter
the repository or linger in the working tree; the bundle travels to the GPU box with the snapshot like any other build
output and is unpacked into a temporary directory while a test or the bench runs).  With them, `oracle/ref_js/run_ref.js` times the reference's OWN `Point.multiply`,
`multiplyUnsafe` and `pippenger` on the box's host cores (bench.py `cpu_baseline.kind = "reference"`) and produces
known answers that pin the Python oracle a second time (tests/test_reference_js.py).

What it does - a tokenizer and one structural pass, no type checking, nothing semantic is rewritten:
  * `import type`, `type` specifiers, type aliases, interfaces, overload signatures, abstract members: removed
  * `: T` annotations on parameters, variables, class fields and return types (incl. `x is T` predicates): removed
  * generic parameter lists and call-site type arguments `<...>`: removed
  * `as T`, `as const`, `satisfies T`, postfix `!`, `implements ...`, access modifiers / readonly / abstract: removed
  * import specifiers `./x.ts` -> `./x.mjs`, `@noble/hashes/*.js` -> the local shim (oracle/ref_js/hashes_shim.mjs:
    node:crypto SHA-2 / HMAC and the few byte helpers the path uses)
  * `a?.b` / `a ?? b` (a handful of places) -> conditional expressions
Everything else - every line of arithmetic - is the reference's text, byte for byte.

    python oracle/ref_js/downlevel.py [--src /root/reference/src] --out <dir> [files...]      (normally through oracle/refjs.py build())
"""
import argparse
import os
import re
import sys

KEYWORDS_CTRL = {"if", "for", "while", "switch", "catch", "with", "return", "typeof", "instanceof", "in", "of", "new", "delete",
                 "void", "throw", "case", "do", "else", "yield", "await"}
TS_MEMBER_MODS = {"public", "private", "protected", "readonly", "abstract", "override", "declare"}
PUNCT3 = ["...", "===", "!==", "**=", "&&=", "||=", "??="]   # no "<<": `TRet<<T>(x: T) => T>` opens two type lists
PUNCT2 = ["=>", "==", "!=", "<=", "&&", "||", "??", "?.", "++", "--", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "**"]


class Tok:
    __slots__ = ("k", "t", "s", "e", "ws", "nl")

    def __init__(self, k, t, s, e, ws, nl):
        self.k, self.t, self.s, self.e, self.ws, self.nl = k, t, s, e, ws, nl

    def __repr__(self):
        return "%s:%r" % (self.k, self.t)


def tokenize(src):
    toks, i, n = [], 0, len(src)
    ws = nl = False

    def prev_allows_regex():
        if not toks:
            return True
        p = toks[-1]
        if p.k in ("num", "str", "tpl", "regex"):
            return False
        if p.k == "id":
            return p.t in KEYWORDS_CTRL
        if p.t == "!" and not p.ws and len(toks) > 1 and (toks[-2].k == "id" or toks[-2].t in (")", "]")):
            return False      # postfix non-null assertion: `x! / 2` is a division
        return p.t not in (")", "]", "}")
    while i < n:
        c = src[i]
        if c in " \t\r":
            i += 1
            ws = True
            continue
        if c == "\n":
            i += 1
            ws = nl = True
            continue
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
            ws = True
            continue
        if src.startswith("/*", i):
            j = src.find("*/", i + 2)
            if "\n" in src[i:j]:
                nl = True
            i = j + 2
            ws = True
            continue
        s = i
        if c.isalpha() or c in "_$#":
            i += 1
            while i < n and (src[i].isalnum() or src[i] in "_$"):
                i += 1
            k = "id"
        elif c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            m = re.compile(r"0[xX][0-9a-fA-F_]+n?|0[bB][01_]+n?|0[oO][0-7_]+n?|(\d[\d_]*\.?[\d_]*|\.\d[\d_]*)([eE][+-]?\d+)?n?").match(src, i)
            i = m.end()
            k = "num"
        elif c in "'\"":
            i += 1
            while src[i] != c:
                i += 2 if src[i] == "\\" else 1
            i += 1
            k = "str"
        elif c == "`":
            i += 1
            depth = 0
            while True:
                ch = src[i]
                if ch == "\\":
                    i += 2
                    continue
                if depth == 0 and ch == "`":
                    i += 1
                    break
                if ch == "$" and src[i + 1] == "{":
                    depth += 1
                    i += 2
                    continue
                if depth and ch == "{":
                    depth += 1
                elif depth and ch == "}":
                    depth -= 1
                i += 1
            k = "tpl"
        elif c == "/" and prev_allows_regex():
            i += 1
            incls = False
            while True:
                ch = src[i]
                if ch == "\\":
                    i += 2
                    continue
                if ch == "[":
                    incls = True
                elif ch == "]":
                    incls = False
                elif ch == "/" and not incls:
                    i += 1
                    break
                i += 1
            while i < n and src[i].isalpha():
                i += 1
            k = "regex"
        else:
            k = "p"
            for cand in PUNCT3:
                if src.startswith(cand, i):
                    i += 3
                    break
            else:
                for cand in PUNCT2:
                    if src.startswith(cand, i):
                        i += 2
                        break
                else:
                    i += 1
        toks.append(Tok(k, src[s:i], s, i, ws, nl))
        ws = nl = False
    toks.append(Tok("eof", "", n, n, True, True))
    return toks


class Stripper:
    def __init__(self, src, name):
        self.src, self.name = src, name
        self.T = tokenize(src)
        self.match = {}
        st = []
        for i, t in enumerate(self.T):
            if t.k == "p" and t.t in "([{":
                st.append(i)
            elif t.k == "p" and t.t in ")]}":
                o = st.pop()
                self.match[o] = i
                self.match[i] = o
        self.cut = []      # (start, end, replacement)
        self.params = set()
        self.class_body = set()
        self.skip_to = -1

    # ---- helpers
    def tt(self, i):
        return self.T[i].t if 0 <= i < len(self.T) else ""

    def is_id(self, i):
        return 0 <= i < len(self.T) and self.T[i].k == "id"

    def delete(self, i, j, repl=""):
        """delete tokens i .. j-1 (source span from token i's start to token j's start, trailing space kept out)"""
        if j <= i:
            return
        s, e = self.T[i].s, self.T[j - 1].e
        self.cut.append((s, e, repl))

    def err(self, i, msg):
        line = self.src.count("\n", 0, self.T[i].s) + 1
        raise SyntaxError("%s:%d: %s near %r" % (self.name, line, msg, self.src[self.T[i].s:self.T[i].s + 60]))

    # ---- type scanner: returns the index of the first token after the type that starts at i
    def angle(self, i):
        """i at '<': index after the matching '>' (type-argument / type-parameter list), or -1"""
        assert self.tt(i) == "<"
        i += 1
        while True:
            t = self.tt(i)
            if t == ">":
                return i + 1
            if t == ",":
                i += 1
                continue
            if t == "" or t in (";", ")"):
                return -1
            # type parameter forms: [const] T [extends U] [= D]  /  type argument forms: any type
            if t == "const" and self.is_id(i + 1):
                i += 1
            j = self.scan_type(i, in_angle=True)
            if j < 0 or j == i:
                return -1
            i = j
            if self.tt(i) == "=" :
                j = self.scan_type(i + 1, in_angle=True)
                if j < 0:
                    return -1
                i = j

    def scan_type(self, i, in_angle=False):
        """-1 if no type can be read at i"""
        if self.tt(i) == "asserts" and self.is_id(i + 1) and self.tt(i + 1) != "is" or (self.tt(i) == "asserts" and self.tt(i + 2) == "is"):
            i += 1
        if (self.is_id(i) or self.tt(i) == "this") and self.tt(i + 1) == "is" and not self.T[i + 1].nl and self.is_id(i + 1):
            nxt = self.tt(i + 2)
            if nxt not in (",", ")", ";", "=", "=>", "{", ""):
                i += 2
        if self.tt(i) in ("|", "&"):
            i += 1
        while True:
            j = self.scan_postfix(i, in_angle)
            if j < 0:
                return -1
            i = j
            if self.tt(i) == "extends" and not self.T[i].nl or (self.tt(i) == "extends"):
                # conditional type  A extends B ? C : D   (or a constraint inside <...>, which has no '?')
                j = self.scan_type(i + 1, in_angle)
                if j < 0:
                    return -1
                if self.tt(j) == "?":
                    a = self.scan_type(j + 1, in_angle)
                    if a < 0 or self.tt(a) != ":":
                        return -1
                    j = self.scan_type(a + 1, in_angle)
                    if j < 0:
                        return -1
                i = j
            if self.tt(i) in ("|", "&"):
                i += 1
                continue
            return i

    def scan_postfix(self, i, in_angle):
        j = self.scan_primary(i, in_angle)
        if j < 0:
            return -1
        while self.tt(j) == "[" and not self.T[j].nl:
            j = self.match[j] + 1
        return j

    def scan_primary(self, i, in_angle):
        t = self.tt(i)
        k = self.T[i].k
        if t == "(":
            c = self.match[i]
            if self.tt(c + 1) == "=>" and self.looks_like_params(i):
                return self.scan_type(c + 2, in_angle)
            return c + 1
        if t == "new" and self.tt(i + 1) in ("(", "<"):
            return self.scan_primary(i + 1, in_angle)
        if t == "abstract" and self.tt(i + 1) == "new":
            return self.scan_primary(i + 1, in_angle)
        if t == "<":   # generic function type
            a = self.angle(i)
            if a < 0:
                return -1
            return self.scan_primary(a, in_angle)
        if t in ("{", "["):
            return self.match[i] + 1
        if t == "typeof":
            j = i + 1
            if self.tt(j) == "import":
                return -1
            if not self.is_id(j):
                return -1
            j += 1
            while self.tt(j) == "." and self.is_id(j + 1):
                j += 2
            if self.tt(j) == "<":
                a = self.angle(j)
                if a > 0:
                    j = a
            return j
        if t in ("keyof", "readonly", "unique", "infer"):
            if t == "infer":
                j = i + 2
                if self.tt(j) == "extends":
                    save = self.scan_postfix(j + 1, in_angle)
                    if save > 0 and self.tt(save) != "?":
                        return save
                return j
            return self.scan_postfix(i + 1, in_angle)
        if k in ("str", "num", "tpl"):
            return i + 1
        if t == "-" and self.T[i + 1].k == "num":
            return i + 2
        if k == "id":
            if t in ("as", "is", "satisfies", "extends", "implements", "in", "of", "instanceof", "return") and False:
                return -1
            j = i + 1
            while self.tt(j) == "." and self.is_id(j + 1):
                j += 2
            if self.tt(j) == "<" and not (self.T[j].ws and not self.T[j + 1].ws and False):
                a = self.angle(j)
                if a > 0:
                    j = a
                elif in_angle:
                    return -1
            return j
        return -1

    def looks_like_params(self, o):
        """inside a TYPE: is `( ... )` at o a parameter list (function type) rather than a parenthesised type?"""
        f = o + 1
        t = self.tt(f)
        if t in (")", "..."):
            return True
        if t in ("{", "["):
            return self.tt(self.match[f] + 1) in (":", ",", ")", "=")
        if self.is_id(f):
            return self.tt(f + 1) in (":", ",", ")", "?")
        return False

    # ---- parameter lists
    def strip_params(self, o):
        """o: index of '(' of a parameter list: remove `?` / `: type` / accessibility modifiers of every parameter"""
        c = self.match[o]
        i = o + 1
        while i < c:
            # one parameter: [modifiers] [...]binding[?] [: type] [= default]
            while self.tt(i) in TS_MEMBER_MODS and (self.is_id(i + 1) or self.tt(i + 1) in ("{", "[", "...")):
                self.delete(i, i + 1)
                i += 1
            if self.tt(i) == "...":
                i += 1
            if self.tt(i) == "this" and self.tt(i + 1) == ":":      # `this` parameter: erased entirely
                j = self.scan_type(i + 2)
                if self.tt(j) == ",":
                    j += 1
                self.delete(i, j)
                i = j
                continue
            if self.tt(i) in ("{", "["):
                self.walk(i + 1, self.match[i])
                i = self.match[i] + 1
            elif self.is_id(i):
                i += 1
            else:
                self.err(i, "parameter expected")
            if self.tt(i) == "?":
                self.delete(i, i + 1)
                i += 1
            if self.tt(i) == ":":
                j = self.scan_type(i + 1)
                if j < 0:
                    self.err(i, "cannot read parameter type")
                self.delete(i, j)
                i = j
            if self.tt(i) == "=":
                j = i + 1
                depth_end = c
                # default value: up to the next top-level comma
                k = j
                while k < depth_end and self.tt(k) != ",":
                    k = self.match[k] + 1 if self.tt(k) in ("(", "[", "{") and self.T[k].k == "p" else k + 1
                self.walk(j, k)
                i = k
            if self.tt(i) == ",":
                i += 1
            elif i < c:
                self.err(i, "',' expected in parameter list")

    def after_params(self, c):
        """c: index of ')' of a parameter list: strip the return type; returns the index of the next token"""
        i = c + 1
        if self.tt(i) == ":":
            j = self.scan_type(i + 1)
            if j < 0:
                self.err(i, "cannot read return type")
            self.delete(i, j)
            i = j
        return i

    def is_params(self, o):
        """is the '(' at o the parameter list of an arrow function / function / method?"""
        c = self.match[o]
        n = self.tt(c + 1)
        p = self.tt(o - 1)
        pk = self.T[o - 1].k if o > 0 else ""
        if n == "=>":
            return True
        named = pk == "id" and p not in KEYWORDS_CTRL and p not in ("super",)
        if n == ":":
            j = self.scan_type(c + 2)
            if j > 0 and self.tt(j) == "=>":
                # `cond ? (a) : b => c` is not something the reference writes
                return True
            if j > 0 and self.tt(j) == "{" and (named or p == "function" or p == ">" or p == "*"):
                return True
            if j > 0 and self.tt(j) == ";" and (named or p == ">") and self.in_class_member(o):
                return True
            return False
        if n == "{" and (named or p == "function" or p == "*") and p not in ("await",):
            # method / function definition (a call cannot be followed by a block)
            return not (pk == "id" and p in KEYWORDS_CTRL)
        return False

    def in_class_member(self, o):
        return False

    # ---- the walk
    def walk(self, i, end):
        T = self.T
        while i < end:
            t = T[i]
            tx = t.t
            if i < self.skip_to:
                i += 1
                continue
            if t.k == "id":
                prev = self.tt(i - 1)
                stmt_start = i == 0 or prev in (";", "{", "}", "") or (T[i].nl and prev not in (".", "=", "(", ",", "?", ":", "=>", "&&", "||", "+", "-", "*", "return", "export"))
                if tx == "import" and self.tt(i + 1) != "(" and self.tt(i + 1) != ".":
                    i = self.do_import(i)
                    continue
                if tx == "export":
                    j = self.do_export(i)
                    if j is not None:
                        i = j
                        continue
                    i += 1
                    continue
                if tx in ("type", "interface") and stmt_start and self.is_id(i + 1) and not T[i + 1].nl and self.tt(i + 2) in ("=", "<", "{", "extends"):
                    i = self.do_type_decl(i, i)
                    continue
                if tx == "declare" and stmt_start and self.is_id(i + 1):
                    j = i
                    while self.tt(j) != ";":
                        j = self.match[j] + 1 if T[j].k == "p" and T[j].t in "([{" else j + 1
                    self.delete(i, j + 1)
                    i = j + 1
                    continue
                if tx == "abstract" and self.tt(i + 1) == "class":
                    self.delete(i, i + 1)
                    i += 1
                    continue
                if tx == "class" and prev != ".":
                    i = self.do_class(i)
                    continue
                if tx == "function" and prev != ".":
                    i = self.do_function(i, i)
                    continue
                if tx in ("const", "let", "var") and prev != "." and (self.is_id(i + 1) or self.tt(i + 1) in ("{", "[")):
                    i = self.do_var(i)
                    continue
                if tx in ("as", "satisfies") and prev != "." and not t.nl and self.expr_end(i - 1):
                    j = i + 1
                    if self.tt(j) == "const":
                        j += 1
                    else:
                        j = self.scan_type(j)
                        if j < 0:
                            self.err(i, "cannot read the type after `%s`" % tx)
                    self.delete(i, j)
                    i = j
                    continue
                # call-site type arguments  f<T>(...)  /  new X<T>(...)
                if self.tt(i + 1) == "<" and tx not in KEYWORDS_CTRL or (self.tt(i + 1) == "<" and tx == "new"):
                    a = self.try_type_args(i + 1)
                    if a > 0:
                        self.delete(i + 1, a)
                        i = a
                        continue
                i += 1
                continue
            if t.k != "p":
                i += 1
                continue
            if tx == "(":
                if self.is_params(i):
                    self.strip_params(i)
                    i = self.after_params(self.match[i])
                    continue
                i += 1
                continue
            if tx == "<":
                # generic arrow function  <T>(x: T) => ...   at an expression start
                prev = self.tt(i - 1)
                if prev in ("=", "(", ",", ":", "?", "return", "=>", "[", "{", "&&", "||", "??", "") or i == 0:
                    a = self.angle(i)
                    if a > 0 and self.tt(a) == "(" and self.is_params(a):
                        self.delete(i, a)
                        i = a
                        continue
                i += 1
                continue
            if tx == ")" and self.tt(i + 1) == "<" and not self.T[i + 1].ws:
                # type arguments on a call RESULT:  getLoop(...)<P>(values)  (fft.ts:558) - only the plain form `<Ident, ...>(`
                a = self.try_type_args(i + 1)
                if a > 0 and all(self.is_id(k) or self.tt(k) in (",", ".") for k in range(i + 2, a - 1)):
                    self.delete(i + 1, a)
                    i = a
                    continue
            if tx == "!" and i > 0 and not t.ws and self.expr_end(i - 1) and self.tt(i + 1) != "=":
                self.delete(i, i + 1)
                i += 1
                continue
            if tx == "?." :
                self.do_optional_chain(i)
                i += 1
                continue
            i += 1
        return i

    def expr_end(self, i):
        t = self.T[i]
        if t.k in ("num", "str", "tpl", "regex"):
            return True
        if t.k == "id":
            return t.t not in KEYWORDS_CTRL or t.t in ("this", "super")
        return t.t in (")", "]", "}")

    def try_type_args(self, i):
        """i at '<' after an identifier: index after '>' if this is a type-argument list followed by a call"""
        a = self.angle(i)
        if a > 0 and self.tt(a) == "(" and not self.T[a].nl:
            return a
        if a > 0 and self.T[a].k == "tpl":
            return a
        return -1

    def do_optional_chain(self, i):
        """a?.b  ->  (a == null ? undefined : a.b)   for `a` a plain member chain and `.b` a property / call chain"""
        T = self.T
        s = i - 1
        while True:
            if T[s].k == "p" and T[s].t in (")", "]"):
                s = self.match[s]
                if self.is_id(s - 1) and T[s].t == "(":
                    s -= 1
            if self.is_id(s) and self.tt(s - 1) == ".":
                s -= 2
                continue
            break
        e = i + 1
        if self.tt(e) in ("(", "["):
            e = self.match[e] + 1
        else:
            e += 1
        while self.tt(e) in ("(", "[") and not T[e].ws:
            e = self.match[e] + 1
        left = self.src[T[s].s:T[i - 1].e]
        right = self.src[T[i].e:T[e - 1].e]
        dot = "" if self.tt(i + 1) in ("(", "[") else "."
        self.cut.append((T[s].s, T[e - 1].e, "(%s == null ? undefined : %s%s%s)" % (left, left, dot, right)))
        self.skip_to = e

    def do_import(self, i):
        T = self.T
        j = i
        while self.tt(j) != ";" and not (T[j].k == "str" and (T[j + 1].nl or self.tt(j + 1) == ";")):
            j += 1
        if T[j].k == "str" and self.tt(j + 1) == ";":
            j += 1
        end = j + 1
        if self.tt(i + 1) == "type":
            self.delete(i, end)
            return end
        self.import_specs(i, end)
        return end

    def import_specs(self, i, end):
        T = self.T
        for k in range(i, end):
            if T[k].k == "p" and T[k].t == "{":
                c = self.match[k]
                items, cur = [], k + 1
                m = k + 1
                while m <= c:
                    if self.tt(m) == "," or m == c:
                        if m > cur:
                            items.append((cur, m))
                        cur = m + 1
                    m += 1
                kept = [self.src[T[a].s:T[b - 1].e] for a, b in items if self.tt(a) != "type"]
                self.cut.append((T[k].s, T[c].e, "{ " + ", ".join(kept) + " }"))
            if T[k].k == "str":
                self.cut.append((T[k].s, T[k].e, map_path(T[k].t)))

    def do_export(self, i):
        nxt = self.tt(i + 1)
        if nxt in ("type", "interface") and (self.is_id(i + 2) or self.tt(i + 2) == "{"):
            if self.tt(i + 2) == "{":     # export type { A, B } [from '...'];
                j = self.match[i + 2] + 1
                while self.tt(j) != ";":
                    j += 1
                self.delete(i, j + 1)
                return j + 1
            return self.do_type_decl(i, i + 1)
        if nxt == "declare":
            j = i
            while self.tt(j) != ";":
                j = self.match[j] + 1 if self.T[j].k == "p" and self.T[j].t in "([{" else j + 1
            self.delete(i, j + 1)
            return j + 1
        if nxt in ("{", "*"):
            j = i
            while self.tt(j) != ";":
                j += 1
            self.import_specs(i, j + 1)
            return j + 1
        if nxt == "function" or (nxt == "async" and self.tt(i + 2) == "function"):
            return self.do_function(i, i + 1 if nxt == "function" else i + 2)
        return None

    def do_type_decl(self, start, i):
        """`type X = ...;` or `interface X {...}` beginning at token `start` (possibly `export`), keyword at i"""
        T = self.T
        if self.tt(i) == "interface":
            j = i
            while self.tt(j) != "{":
                j += 1
            j = self.match[j] + 1
            self.delete(start, j)
            return j
        j = i + 2
        if self.tt(j) == "<":
            j = self.angle(j)
        if self.tt(j) != "=":
            self.err(i, "type alias without '='")
        k = self.scan_type(j + 1)
        if k < 0:
            self.err(i, "cannot read the aliased type")
        if self.tt(k) == ";":
            k += 1
        elif not self.T[k].nl:
            self.err(k, "type alias does not end at a statement boundary")
        self.delete(start, k)
        return k

    def do_function(self, start, i):
        """`function` keyword at i (statement may begin at `start`: export / async)"""
        j = i + 1
        if self.tt(j) == "*":
            j += 1
        if self.is_id(j) and self.tt(j) not in ("<",):
            j += 1
        if self.tt(j) == "<":
            a = self.angle(j)
            if a < 0:
                self.err(j, "cannot read type parameters")
            self.delete(j, a)
            j = a
        if self.tt(j) != "(":
            self.err(j, "'(' expected after function name")
        c = self.match[j]
        self.strip_params(j)
        k = self.after_params(c)
        if self.tt(k) == ";" or self.tt(k) != "{":      # overload signature: no body
            if self.tt(k) == ";":
                k += 1
            # drop the cuts made inside and remove the whole declaration
            s0 = self.T[start].s
            self.cut = [x for x in self.cut if x[0] < s0]
            self.delete(start, k)
            return k
        return k

    def do_var(self, i):
        j = i + 1
        while True:
            if self.tt(j) in ("{", "["):
                self.walk(j + 1, self.match[j])
                j = self.match[j] + 1
            elif self.is_id(j):
                j += 1
            else:
                return j
            if self.tt(j) == "!" :
                self.delete(j, j + 1)
                j += 1
            if self.tt(j) == ":":
                k = self.scan_type(j + 1)
                if k < 0:
                    self.err(j, "cannot read variable type")
                self.delete(j, k)
                j = k
            if self.tt(j) == "," and (self.is_id(j + 1)) and self.tt(j + 2) in (":", ",", ";", "=") and self.T[j + 1].t not in KEYWORDS_CTRL:
                j += 1
                continue
            return j

    def do_class(self, i):
        T = self.T
        j = i + 1
        if self.is_id(j) and self.tt(j) not in ("extends", "implements"):
            j += 1
        if self.tt(j) == "<":
            a = self.angle(j)
            self.delete(j, a)
            j = a
        if self.tt(j) == "extends":
            j += 1
            k = j
            while self.tt(k) not in ("{", "implements", "<"):
                k = self.match[k] + 1 if T[k].k == "p" and T[k].t in "([" else k + 1
            self.walk(j, k)
            j = k
            if self.tt(j) == "<":
                a = self.angle(j)
                self.delete(j, a)
                j = a
        if self.tt(j) == "implements":
            k = j
            while self.tt(k) != "{":
                k += 1
            self.delete(j, k)
            j = k
        if self.tt(j) != "{":
            self.err(j, "class body expected")
        c = self.match[j]
        self.class_members(j + 1, c)
        return c + 1

    def class_members(self, i, end):
        T = self.T
        while i < end:
            if self.tt(i) == ";":
                i += 1
                continue
            start = i
            is_abstract = False
            # modifiers
            while True:
                t = self.tt(i)
                nxt = self.tt(i + 1)
                is_name_pos = nxt in ("(", "<", ":", "=", ";", "?", "!", "}") or T[i + 1].nl
                if t in TS_MEMBER_MODS and not is_name_pos:
                    if t == "abstract":
                        is_abstract = True
                    self.delete(i, i + 1)
                    i += 1
                    continue
                if t in ("static", "async", "get", "set") and not is_name_pos:
                    i += 1
                    if t == "static" and self.tt(i) == "{":      # static initialisation block
                        self.walk(i + 1, self.match[i])
                        i = self.match[i] + 1
                        t = None
                        break
                    continue
                break
            if t is None:
                continue
            if self.tt(i) == "*":
                i += 1
            # member name
            if self.tt(i) == "[":
                if self.tt(self.match[i] + 1) == ":" and self.tt(i + 2) == ":":   # index signature
                    k = i
                    while self.tt(k) != ";":
                        k = self.match[k] + 1 if T[k].k == "p" and T[k].t in "([{" else k + 1
                    self.cut = [x for x in self.cut if x[0] < T[start].s]
                    self.delete(start, k + 1)
                    i = k + 1
                    continue
                self.walk(i + 1, self.match[i])
                i = self.match[i] + 1
            elif T[i].k in ("id", "str", "num"):
                i += 1
            else:
                self.err(i, "class member name expected")
            if self.tt(i) in ("?", "!"):
                self.delete(i, i + 1)
                i += 1
            if self.tt(i) == "<":
                a = self.angle(i)
                self.delete(i, a)
                i = a
            if self.tt(i) == "(":
                c = self.match[i]
                self.strip_params(i)
                k = self.after_params(c)
                if self.tt(k) == "{":
                    self.walk(k + 1, self.match[k])
                    i = self.match[k] + 1
                    continue
                # signature without a body: abstract member or overload
                if self.tt(k) == ";":
                    k += 1
                self.cut = [x for x in self.cut if x[0] < T[start].s]
                self.delete(start, k)
                i = k
                continue
            # field
            if self.tt(i) == ":":
                k = self.scan_type(i + 1)
                if k < 0:
                    self.err(i, "cannot read field type")
                self.delete(i, k)
                i = k
            if is_abstract:
                k = i
                while self.tt(k) != ";":
                    k += 1
                self.cut = [x for x in self.cut if x[0] < T[start].s]
                self.delete(start, k + 1)
                i = k + 1
                continue
            if self.tt(i) == "=":
                k = i + 1
                while k < end and self.tt(k) != ";" and not (T[k].nl and self.expr_end(k - 1) and self.tt(k) not in (".", "?", ":", "+", "-", "*", "&&", "||")):
                    k = self.match[k] + 1 if T[k].k == "p" and T[k].t in "([{" else k + 1
                self.walk(i + 1, k)
                i = k
            if self.tt(i) == ";":
                i += 1
            elif not T[i].nl and i < end:
                self.err(i, "';' expected after class field")

    def run(self):
        self.walk(0, len(self.T) - 1)
        out, pos = [], 0
        for s, e, r in sorted(self.cut):
            if s < pos:
                if e <= pos:
                    continue
                raise SyntaxError("%s: overlapping edits at %d" % (self.name, s))
            out.append(self.src[pos:s])
            if not r and s > 0 and e < len(self.src) and (self.src[s - 1].isalnum() or self.src[s - 1] in "_$") and (self.src[e].isalnum() or self.src[e] in "_$"):
                r = " "
            out.append(r)
            pos = e
        out.append(self.src[pos:])
        return "".join(out)


SHIM_PREFIX = "./"
# packages the reference's TEST and BENCHMARK files import that are not installed here -> the stand-ins of oracle/ref_js/harness/
# (a test runner, a property-test generator, a benchmark loop, the gz reader of an absent vectors submodule): harness, no arithmetic
HARNESS = {"@paulmillr/jsbt/test.js": "harness/jsbt_test.mjs", "@paulmillr/jsbt/benchmark.js": "harness/jsbt_bench.mjs",
           "@paulmillr/jsbt/benchmark-compare.js": "harness/jsbt_compare.mjs", "fast-check": "harness/fast_check.mjs",
           "./vectors/acvp-vectors/utils.js": "harness/acvp_utils.mjs"}


def map_path(lit):
    q = lit[0]
    p = lit[1:-1]
    if p.startswith("@noble/hashes/"):
        return q + SHIM_PREFIX + "hashes_shim.mjs" + q
    if p in HARNESS:
        return q + SHIM_PREFIX + HARNESS[p] + q
    if p.startswith("@noble/curves/") and p.endswith(".js"):      # the package's own name (test helpers): its src/ tree
        return q + SHIM_PREFIX + "src/" + p[len("@noble/curves/"):-3] + ".mjs" + q
    if p.endswith(".ts"):
        p = p[:-3] + ".mjs"
    return q + p + q


# ---- the GPU redirect: the change a maintainer of the reference would make to src/abstract/curve.ts (INTEGRATION.md shows it as
# a diff).  Written as TypeScript and applied to the reference's text BEFORE the types are stripped, at two anchors: a backend
# registry in front of `pippenger`, and one dispatch line after its argument checks and its empty-input return - so the
# reference's own validation (and its error messages) always runs, inputs below `minPoints` fall through to the reference's loop.
HOOK_REGISTRY = """/**
 * Optional accelerator for {@link pippenger}: a backend registered for a Point constructor receives the (already validated)
 * arguments of every call with at least `minPoints` points; smaller inputs run the loop below.
 */
export type MSMBackend = { minPoints: number; msm: (c: any, points: any[], scalars: bigint[]) => any };
const msmBackends = new WeakMap<object, MSMBackend>();
export function setMSMBackend(c: object, backend?: MSMBackend): void {
  if (backend === undefined) msmBackends.delete(c);
  else msmBackends.set(c, backend);
}
"""
HOOK_DISPATCH = """  const backend = msmBackends.get(c);
  if (backend !== undefined && plength >= backend.minPoints) return backend.msm(c, points, scalars) as P;
"""


def apply_gpu_hook(src):
    a1 = re.search(r"^/\*\*\n(?: \*[^\n]*\n)*? \*/\nexport function pippenger<", src, flags=re.M)
    if a1 is None:
        a1 = re.search(r"^export function pippenger<", src, flags=re.M)
    a2 = re.search(r"^  if \(plength === 0\) return zero as P;\n", src, flags=re.M)
    if a1 is None or a2 is None or a2.start() < a1.start():
        raise SyntaxError("abstract/curve.ts: pippenger anchors not found - the hook needs a look")
    # the doc comment regex is lazy over comment lines, but may start at an EARLIER comment: take the last '/**' before the function
    fn = src.index("export function pippenger<", a1.start())
    doc = src.rfind("/**", 0, fn)
    at = doc if doc >= 0 and src[doc:fn].count("*/") == 1 else fn
    return src[:at] + HOOK_REGISTRY + src[at:a2.end()] + HOOK_DISPATCH + src[a2.end():]


NULLISH = re.compile(r"\?\?=?")


def post(text, relname):
    """the few ES2020+ operators left (none in the arithmetic): `a ?? b` with plain operands"""
    def repl(m):
        raise SyntaxError("%s: `??` needs a hand patch" % relname)
    # JSON modules (`import x from './a.json' with { type: 'json' }`): Node 12 has neither import attributes nor JSON modules -
    # the packer (oracle/refjs.py) writes `a.json.mjs` (`export default <the JSON>`) beside the data file
    text = re.sub(r"(import\s+\w+\s+from\s+'[^']*\.json)'\s*with\s*\{[^}]*\}", r"\1.mjs'", text)
    if "??" in re.sub(r"//[^\n]*|/\*.*?\*/|'[^'\n]*'|\"[^\"\n]*\"|`[^`]*`", "", text, flags=re.S):
        text = re.sub(r"([A-Za-z_$][\w$.]*)\s*\?\?\s*([A-Za-z_$][\w$.]*|'[^']*'|\d+n?)", r"(\1 != null ? \1 : \2)", text)
        # a parenthesised left operand (what `a?.b ?? c()` has become): member reads only, so evaluating it twice is harmless
        text = re.sub(r"(\((?:[^()]|\([^()]*\))*\))\s*\?\?\s*([A-Za-z_$][\w$]*(?:\.[\w$]+|\(\))*)", r"(\1 != null ? \1 : \2)", text)
    return text


DEFAULT_FILES = ["utils.ts", "abstract/modular.ts", "abstract/curve.ts", "abstract/weierstrass.ts", "abstract/der.ts",
                 "abstract/edwards.ts", "abstract/hash-to-curve.ts", "abstract/tower.ts", "abstract/bls.ts", "abstract/fft.ts",
                 "abstract/montgomery.ts", "abstract/frost.ts", "abstract/oprf.ts", "secp256k1.ts", "ed25519.ts", "bls12-381.ts"]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default="/root/reference/src")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(here), "_ref", "js"))
    ap.add_argument("--gpu-hook", action="store_true", help="apply the MSM-backend patch to abstract/curve.ts (the redirect of INTEGRATION.md)")
    ap.add_argument("--print-hook-diff", action="store_true", help="print the patch as a unified diff against the reference's file and exit")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    if a.print_hook_diff:
        import difflib
        rel = "abstract/curve.ts" if os.path.exists(os.path.join(a.src, "abstract/curve.ts")) else "src/abstract/curve.ts"
        old = open(os.path.join(a.src, rel)).read()
        sys.stdout.writelines(difflib.unified_diff(old.splitlines(True), apply_gpu_hook(old).splitlines(True), "a/src/abstract/curve.ts",
                                                   "b/src/abstract/curve.ts", n=2))
        return
    files = a.files or DEFAULT_FILES
    ok = True
    for rel in files:
        src = open(os.path.join(a.src, rel)).read()
        global SHIM_PREFIX
        SHIM_PREFIX = "../" * rel.count("/") or "./"
        try:
            if a.gpu_hook and rel.endswith("abstract/curve.ts"):
                src = apply_gpu_hook(src)
            text = post(Stripper(src, rel).run(), rel)
        except SyntaxError as e:
            print("FAIL", e, file=sys.stderr)
            ok = False
            continue
        dst = os.path.join(a.out, rel[:-3] + ".mjs")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write("// DERIVED from the reference's src/%s by oracle/ref_js/downlevel.py (types stripped) - test infrastructure, not committed\n" % rel)
            f.write(text)
        print("ok  ", rel, "->", os.path.relpath(dst))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
