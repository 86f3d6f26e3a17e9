"""Import-compatibility shim: `pnpflow.*` names of the reference's restoration path resolve to the
MI355X engine in `pnpflow_amd`.  Nothing else of the reference package exists here."""
