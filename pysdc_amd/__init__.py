"""MI355X-native SDC sweep engine behind pySDC's sweeper/problem plug-in API."""

__version__ = '0.1.0'
