"""VSS (IEEE Very Small Size) environments served by the MI355X step engine."""
