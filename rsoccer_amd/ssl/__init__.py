"""SSL (RoboCup Small Size League) environments served by the MI355X step engine."""
