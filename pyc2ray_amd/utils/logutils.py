"""printlog, as pyc2ray/utils/logutils.py:1-15."""

__all__ = ["printlog", "printlog_lines"]


def printlog(s, filename, quiet=False, end='\n'):
    """Append `s` to the log file and, unless quiet, print it."""
    if filename is not None:
        with open(filename, "a") as f:
            f.write(s + end)
    if not quiet:
        print(s, end=end)


def printlog_lines(lines, filename, quiet=False):
    """The same output as one printlog call per (text, end) pair of `lines`, with the log file opened once: the loop of a
    time step reports five lines per outer iteration, and opening the file for each of them is what took the time."""
    if filename is not None and lines:
        with open(filename, "a") as f:
            f.write("".join(s + end for s, end in lines))
    if not quiet:
        for s, end in lines:
            print(s, end=end)
