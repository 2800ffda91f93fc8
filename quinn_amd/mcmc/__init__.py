"""Mirror of the reference package of the same name (hot-path members only)."""
