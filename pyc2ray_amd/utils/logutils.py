"""printlog, as pyc2ray/utils/logutils.py:1-15."""

__all__ = ["printlog"]


def printlog(s, filename, quiet=False, end='\n'):
    """Append `s` to the log file and, unless quiet, print it."""
    if filename is not None:
        with open(filename, "a") as f:
            f.write(s + end)
    if not quiet:
        print(s, end=end)
