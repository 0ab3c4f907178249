"""Version of the MI355X-native MTM drop-in (the reference is at 2.0.1, MTM/version.py:5)."""
__version__ = "2.0.1+mi355x.2"
