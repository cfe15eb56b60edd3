"""Static rules of the hand-written kernels, checked on the sources (CPU suite, no compiler needed).

Rule (ADVICE r2 / VERDICT r2 item 6): a cross-lane operation -- DPP wave shift or row rotation, v_readlane / v_readfirstlane, ds_bpermute
shuffle, permlane swap, wavefront vote, barrier -- executes with the EXEC mask of its control flow.  Inside a block some lanes of the wavefront
have not entered (a lane-dependent `if`, a loop whose trip count or exits depend on the lane) it reads from lanes that are switched off:
undefined at SOURCE level, whatever the compiler does.  So:
  1. the per-lane phase functions (cclqr_chain.h, cclqr_dev.h, cclqr_lin_dev.h, cclqr_loop.h: called under lane predicates) contain NO
     cross-lane operation at all;
  2. in the kernel bodies and the Newton orchestration every cross-lane operation sits under control flow whose every enclosing condition is
     on a list of wavefront-uniform expressions (launch arguments, mechanism constants, loop counters with uniform bounds, wavefront votes).
     A new enclosing condition fails this test until a human has looked at it and added it to the list;
  3. no loop in device code has more than one exit (`break` / `return` inside a loop body) unless the exit condition is a wavefront vote
     or on the uniform list.
The rule is lexical (brace matching on the source text): a tripwire, not a proof."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc")

CROSS = re.compile(r"\b(lane_read|from_lane\s*<|child_sum\s*<|tr_other_half|wave_from_prev|wave_from_next|from_prev\s*<|from_next\s*<|group_sum\s*<|dpp_f64\s*<|dpp_row_ror\s*<|wave_max_key|key_dpp_max\s*<|other_half|__shfl\w*|"
                   r"__builtin_amdgcn_(readlane|readfirstlane|update_dpp|mov_dpp|permlane\w*|ds_bpermute|ds_swizzle)|__syncthreads|__any|__all|__ballot)\b")
# definitions of the cross-lane helpers themselves (their bodies ARE the cross-lane instruction)
HELPER_DEF = re.compile(r"(double|void|int|v4d|unsigned long long)\s+(lane_read|from_lane|child_sum|tr_other_half|wave_from_prev|wave_from_next|from_prev|from_next|group_sum|dpp_f64|dpp_row_ror|wave_max_key|key_dpp_max|other_half)\s*\(")

PER_LANE_HEADERS = ["cclqr_chain.h", "cclqr_dev.h", "cclqr_lin_dev.h", "cclqr_loop.h", "cclqr_treereg.h"]
ORCHESTRATION = ["rollout_chain.hip", "rollout_treereg.hip", "rollout_loop.hip", "linearize.hip", "cclqr_newton.h"]

# wavefront-uniform conditions that may enclose a cross-lane operation (regexes on the condition text, whitespace collapsed)
UNIFORM = [r"^!?__any\(", r"^!?__all\(", r"^kk < nsteps$", r"^iter <= NEWTON_MAXIT$", r"^lv <= LINE_MAXIT", r"^ls <= LINE_MAXIT", r"^ci < nchains$", r"^c < nchains$",
           r"^i < P\.steps$", r"^j < P\.steps$", r"^gate$", r"^CR && cr$", r"^traj_out", r"^EXTRA", r"^JAC$", r"^TREE$", r"^G == \d+$", r"^i < C->mu$",
           r"^C->", r"^a\.", r"^ap->", r"^M->", r"^NL", r"^l >= 0$", r"^l < nb$", r"^int l = ", r"^int i = 0; i < C->mu", r"^int kk = 0; kk < nsteps",
           r"^int iter = 1; iter <= NEWTON_MAXIT", r"^int lv = 1; lv <= LINE_MAXIT", r"^int ls = 0; ls <= LINE_MAXIT", r"^int ci = 0; ci < nchains", r"^int c = 0; c < nchains",
           r"^int i = 0; i < P\.steps", r"^int j = 0; j < P\.steps", r"^int q = 0; q < (LEVEL_SLOTS|NL)", r"^gate && ", r"^k < ", r"^step < ", r"^int step = ",
           r"^int p = 0; p < ", r"^p < ", r"^int it = ",
           # rollout_chain.hip line search: `mine` is uniform over a 32-lane group and `other` is the partner group's `mine`, so `mine != other`
           # has the same value in both groups of the wavefront
           # rollout_treereg.hip: schedule lengths and the largest child count are mechanism constants (TreeRegDev), the same in every lane
           r"^int s = 0; s < (ne_steps|nb_steps)", r"^s < (ne_steps|nb_steps)", r"^int k = 0; k < maxchild",
           r"^mine != other$", r"^int i = 0; i < \d+; i\+\+$",        # (compile-time unrolled component loops)
           # control phase (round 4): mu = C->mu inputs in chunks of the compile-time CH; the gain table's presence is a controller constant
           r"^int i0 = 0; i0 < mu", r"^int j = 0; j < CH; j\+\+$", r"^int i = 0; i < mu",
           # rollout_loop.hip project_model_kernel: tree reductions over a fixed workgroup size, pivot steps up to the kernel argument ml, and the
           # stop on s_pi -- a __shared__ word written by thread 0 and read by everyone behind a barrier
           r"^int o = PROJ_THREADS / 2; o > 0; o >>= 1$", r"^int k = 0; k < ml", r"^s_pi < 0$",
           # rollout_loop.hip loop_solve: mr = 5 M->nj; `rank` and the pivot it stops on come out of wave_max_key (v_readlane 63: the same in every lane)
           r"^int k = 0; k < mr", r"^int k = rank - 1; k >= 0", r"^it <= ", r"^nsteps", r"^steps", r"^mode", r"^int kk = ", r"^rank < ", r"^int e = t; e < Y\.total",
           # rollout_chain.hip chain_eval (round 5): KL = lanes per link, a template parameter
           r"^JAC && KL > 1$", r"^KL (>|==) \d$"]


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), txt, flags=re.S)
    return re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), txt)


def enclosing_headers(txt, pos):
    """conditions of the control statements whose braces enclose position pos, innermost first (up to the enclosing function body)"""
    out = []
    depth = 0
    i = pos
    while i > 0:
        i -= 1
        ch = txt[i]
        if ch == "}":
            depth += 1
        elif ch == "{":
            if depth > 0:
                depth -= 1
                continue
            # an enclosing block opens here: find what precedes it
            j = i - 1
            while j >= 0 and txt[j].isspace():
                j -= 1
            if txt[j] == ")":                       # `if (...) {`, `for (...) {`, a function head, a lambda
                d2, k = 0, j
                while k >= 0:
                    if txt[k] == ")":
                        d2 += 1
                    elif txt[k] == "(":
                        d2 -= 1
                        if d2 == 0:
                            break
                    k -= 1
                cond = " ".join(txt[k + 1:j].split())
                m = re.search(r"(\w+)\s*$", txt[:k])
                kw = m.group(1) if m else ""
                if kw in ("if", "for", "while", "switch"):
                    out.append((kw, cond))
                else:
                    return out                      # function (or kernel) head: stop
            else:
                m = re.search(r"(\w+)\s*$", txt[:j + 1])
                kw = m.group(1) if m else ""
                if kw in ("else", "do"):
                    out.append((kw, ""))            # the matching `if` condition is checked where the `if` block is scanned
                # a bare block / struct / namespace: keep walking outwards
    return out


def test_phase_functions_hold_no_cross_lane_operation():
    for f in PER_LANE_HEADERS:
        txt = strip_comments(open(os.path.join(CSRC, f)).read())
        hits = [m.group(0) for m in CROSS.finditer(txt)]
        assert not hits, (f, hits[:5])


def test_cross_lane_operations_sit_under_uniform_control_flow():
    seen = 0
    for f in ORCHESTRATION:
        txt = strip_comments(open(os.path.join(CSRC, f)).read())
        for m in CROSS.finditer(txt):
            line_start = txt.rfind("\n", 0, m.start()) + 1
            if HELPER_DEF.search(txt[line_start:m.end() + 40]) and "(" in txt[m.end() - 1:m.end() + 2]:
                continue
            # inside the body of a cross-lane helper definition: the helper is the instruction
            head = txt[max(0, m.start() - 1500):m.start()]
            defs = list(HELPER_DEF.finditer(head))
            if defs and head[defs[-1].end():].count("{") > head[defs[-1].end():].count("}"):
                continue
            seen += 1
            for kw, cond in enclosing_headers(txt, m.start()):
                if kw in ("else", "do"):
                    continue
                ok = any(re.search(p, cond) for p in UNIFORM)
                line = txt.count("\n", 0, m.start()) + 1
                assert ok, "%s:%d: cross-lane operation `%s` under `%s (%s)`: is that condition wavefront-uniform? (tests/test_static_kernel_rules.py)" % (f, line, m.group(0), kw, cond)
    assert seen > 30


def test_device_loops_have_one_exit():
    """`break` in device code only on a wavefront vote (or in host-side table builders); no `return` inside a loop of a kernel body"""
    for f in PER_LANE_HEADERS + ORCHESTRATION + ["cclqr_newton.h"]:
        txt = strip_comments(open(os.path.join(CSRC, f)).read())
        for m in re.finditer(r"\bbreak\s*;", txt):
            stmt_start = max(txt.rfind(";", 0, m.start()), txt.rfind("{", 0, m.start()), txt.rfind("}", 0, m.start())) + 1
            stmt = " ".join(txt[stmt_start:m.end()].split())
            line = txt.count("\n", 0, m.start()) + 1
            if re.match(r"(case\b|default\b)", stmt) or "switch" in " ".join(c for k, c in enclosing_headers(txt, m.start())[:1]):
                continue
            kws = enclosing_headers(txt, m.start())
            if kws and kws[0][0] == "switch":
                continue
            assert re.search(r"if \(!__any\(", stmt) or f == "rollout_loop.hip" and re.search(r"best > |normf1 > normf0|s_pi < 0", stmt), \
                "%s:%d: `%s` -- a second loop exit must be a wavefront vote" % (f, line, stmt)


def test_the_tripwire_trips():
    """the scanner flags a DPP shift under a lane predicate and a two-exit loop, and accepts the same code at uniform control flow"""
    bad = "__global__ void k(Args a) {\n  for (int kk = 0; kk < nsteps; kk++) {\n    if (c.live()) { double p = wave_from_prev(x); }\n  }\n}\n"
    m = CROSS.search(bad)
    hdr = enclosing_headers(bad, m.start())
    assert hdr[0] == ("if", "c.live()") and not any(re.search(p, hdr[0][1]) for p in UNIFORM)
    assert any(re.search(p, hdr[1][1]) for p in UNIFORM)
    good = bad.replace("if (c.live()) { double p = wave_from_prev(x); }", "double p = wave_from_prev(x); if (c.live()) { y = p; }")
    m = CROSS.search(good)
    assert all(any(re.search(p, c) for p in UNIFORM) for _, c in enclosing_headers(good, m.start()))
