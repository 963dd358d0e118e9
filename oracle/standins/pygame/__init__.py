"""Stand-in for pygame (only used by the reference's render(), which golden generation never calls)."""
