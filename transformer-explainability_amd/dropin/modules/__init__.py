"""Drop-in for the reference package `modules` (rule libraries)."""
