"""HIP side of the scoring path: C-ABI binding (_capi), tensor wrappers (ops), tower engines (engine)."""
