"""Golden vectors for the driver's LOG FORMAT, made by the reference's own log parsers.

The reference holds no recorded outputs of its kernels, but two of its files are plain Python and run here: parse_time.py and
parse_counter.py (the consumers of the logs main_qgtc.py / cluster_gcn.py write; 0_7a_eval_QGTC_cluster_GCN.py:37-41 and
4_8_zero_tile_jumping.py:44). This script composes logs with the package's OWN format strings (driver.args_line, driver.AVG_EPOCH_FORMAT,
the extension's `counter_global: %d` / `counter: %d` lines, qgtc_torch.cpp:286,298), runs the two reference scripts on them where they lie
under /root/reference and stores log text + the scripts' stdout in parser_golden.json. tests/test_parser_golden.py re-composes the logs
(format drift fails) and checks the driver's own figures against what the reference's parsers printed.

    python tests/golden/make_parser_golden.py        (needs /root/reference; not run by the tests)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"

# (dataset, dim, classes): the tables of 0_7a_eval_QGTC_cluster_GCN.py:12-17 / 4_8_zero_tile_jumping.py:9-13
TIME_RUNS = [("artist", 100, 12, 263.571), ("soc-BlogCatalog", 128, 39, 209.523), ("ppi", 50, 121, 0.018), ("ogbn-arxiv", 128, 40, 0.022)]
# cumulative counter lines of three runs (values as the extension prints them: running totals, one pair per batch)
COUNTER_RUNS = [("Proteins", 29, 2, [120, 260, 420], [30, 50, 90]),
                ("artist", 100, 12, [1000, 2100], [250, 400]),
                ("ogbn-arxiv", 128, 40, [5000, 9000, 15000, 22000], [900, 2100, 2500, 4100])]


def compose_time_log():
    from qgtc_ppopp22_amd import driver
    lines = []
    for data, d, c, ms in TIME_RUNS:
        args = driver.build_parser().parse_args(["--dataset", data, "--dim", str(d), "--n-hidden", "16", "--n-classes", str(c), "--use_QGTC"])
        lines += [driver.args_line(args), driver.extra_flags_line(args), driver.AVG_EPOCH_FORMAT.format(ms)]
    return "\n".join(lines) + "\n"


def compose_counter_log():
    """File order of a run redirected to a file, as the reference's runs leave it (4_8_zero_tile_jumping.py:30 `>> zerotile_jumping.log`):
    the extension's printf lines first (C stdio flushes its buffer as it fills), Python's `print(args)` when the interpreter exits -
    which is why parse_counter.py:11-19 closes a block AT the Namespace line."""
    from qgtc_ppopp22_amd import driver
    lines = []
    for data, d, c, cum_g, cum_c in COUNTER_RUNS:
        args = driver.build_parser().parse_args(["--dataset", data, "--dim", str(d), "--n-hidden", "16", "--n-classes", str(c), "--use_QGTC",
                                                 "--zerotile_jump", "--n-epochs", "1"])
        for g_, c_ in zip(cum_g, cum_c):
            lines += ["counter_global: %d" % g_, "counter: %d" % c_]
        lines += [driver.args_line(args), driver.extra_flags_line(args)]
    return "\n".join(lines) + "\n"


def run_reference(script, log_text):
    path = os.path.join(HERE, "_tmp.log")
    with open(path, "w") as f:
        f.write(log_text)
    try:
        return subprocess.run([sys.executable, os.path.join(REF, script), path], check=True, capture_output=True, text=True).stdout
    finally:
        os.remove(path)


def main():
    t, c = compose_time_log(), compose_counter_log()
    out = {"made_by": "tests/golden/make_parser_golden.py: /root/reference/parse_time.py and parse_counter.py run on the two logs",
           "time_log": t, "parse_time_stdout": run_reference("parse_time.py", t),
           "counter_log": c, "parse_counter_stdout": run_reference("parse_counter.py", c)}
    with open(os.path.join(HERE, "parser_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["parse_time_stdout"])
    print(out["parse_counter_stdout"])


if __name__ == "__main__":
    main()
