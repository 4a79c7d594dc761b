"""nhd_amd - MI355X-native node filter-and-score engine for the NHD scheduler."""
__version__ = "0.1.0"
