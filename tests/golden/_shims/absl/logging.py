"""Stand-in for absl.logging (absent from this image): forwards to stdlib logging.

Used ONLY while generating golden vectors from the reference's importable numpy pieces.
It carries no arithmetic, so it cannot influence any golden value.
"""
import logging as _l

info = _l.info
warning = _l.warning
error = _l.error
debug = _l.debug
