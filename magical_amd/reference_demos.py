"""`magical.try_download_demos` (magical/reference_demos.py:19-53) -- the name and its local contract only.

The reference fetches qxcv/magical-data from GitHub into `dest` and leaves a `.download-done` marker there; a later call
returns at once when the marker exists (reference_demos.py:27-33).  Networking is outside this engine's scope (SURVEY.md
section 8: the hot path is BaseEnv.step), so: a directory that already holds the marker is accepted exactly as the reference
accepts it, anything else raises with the instruction to place the data there by hand.  The demo files themselves are read
by magical_amd.saved_trajectories.load_demos.
"""
import logging
import os

__all__ = ['try_download_demos']

REFERENCE_DEMO_ZIP = 'https://github.com/qxcv/magical-data/archive/master.zip'
DEFAULT_LOCATION = 'demos'
DONE_FILE = '.download-done'


class DownloadError(Exception):
    pass


def try_download_demos(dest=DEFAULT_LOCATION, progress=True):
    """Return if `dest` already holds the reference's download marker; otherwise say how to put the demonstrations there."""
    if os.path.exists(os.path.join(dest, DONE_FILE)):
        logging.info(f"Demonstrations appear to already be in '{dest}'; to force download, delete that directory and try again")
        return
    raise DownloadError(f"magical_amd does not download anything: unpack {REFERENCE_DEMO_ZIP} into '{dest}' and create "
                        f"'{os.path.join(dest, DONE_FILE)}' (what magical.try_download_demos leaves behind), then call again")
