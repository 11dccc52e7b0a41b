"""The driver's log format against tests/golden/parser_golden.json: logs composed with the package's own format strings, and what the
REFERENCE's parse_time.py / parse_counter.py printed when run on them (tests/golden/make_parser_golden.py made the file in the container
that holds /root/reference; nothing here reads the reference)."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _gold():
    with open(os.path.join(HERE, "golden", "parser_golden.json")) as f:
        return json.load(f)


def _maker():
    spec = importlib.util.spec_from_file_location("make_parser_golden", os.path.join(HERE, "golden", "make_parser_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_logs_are_still_what_the_reference_parsers_were_run_on():
    g, m = _gold(), _maker()
    assert m.compose_time_log() == g["time_log"]
    assert m.compose_counter_log() == g["counter_log"]


def test_parse_time_reads_the_dataset_and_the_epoch_time_from_the_driver_lines():
    """parse_time.py:9-17: comma field 2 of the Namespace line, the number of the `Avg. Epoch:` line; :19-20 prints `dataset , ms`."""
    g, m = _gold(), _maker()
    rows = g["parse_time_stdout"].splitlines()
    assert rows[0] == "dataset , Epoch (ms)"
    assert rows[1:] == ["{} , {}".format(d, float("{:.3f}".format(ms))) for d, _, _, ms in m.TIME_RUNS]


def test_parse_counter_prints_the_rows_the_driver_computes():
    """parse_counter.py:10-34 on a zero-tile log in the driver's format: its header and rows are driver.ZEROTILE_HEADER and
    driver.zerotile_row(...)['line'] - the sums of the cumulative lines, ratio = jumping / non-jumping to three places."""
    from qgtc_ppopp22_amd import driver

    g, m = _gold(), _maker()
    rows = g["parse_counter_stdout"].splitlines()
    assert rows[0] == driver.ZEROTILE_HEADER
    assert rows[1:] == [driver.zerotile_row(d, cg, cc)["line"] for d, _, _, cg, cc in m.COUNTER_RUNS]


def test_only_the_namespace_line_carries_the_word_the_parsers_key_on():
    """Both parsers treat ANY line with `dataset` in it as the Namespace line (parse_time.py:10, parse_counter.py:11): the driver's other
    lines must not contain it, and the Namespace line must have the dataset where each script looks (field 2 / field 1, >= 5 fields)."""
    from qgtc_ppopp22_amd import driver

    a = driver.build_parser().parse_args(["--dataset", "ppi", "--use_QGTC"])
    line = driver.args_line(a)
    assert line.split(",")[2].split("=")[1].strip("'") == "ppi" and len(line.split(",")) >= 5      # parse_time.py:11-12
    z = driver.build_parser().parse_args(["--dataset", "ppi", "--use_QGTC", "--zerotile_jump"])
    assert driver.args_line(z).split(",")[1].split("=")[1].strip("'") == "ppi"                      # parse_counter.py:12
    for other in (driver.extra_flags_line(a), driver.AVG_EPOCH_FORMAT.format(1.0)):
        assert "dataset" not in other and "counter" not in other
    assert "Avg. Epoch:" not in driver.extra_flags_line(a)
