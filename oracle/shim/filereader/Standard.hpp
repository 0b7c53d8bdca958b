// TEST SCAFFOLDING (oracle/_ref build): stand-in for rapidgzip's StandardFileReader.
#pragma once
#include <string>
namespace rapidgzip { struct StandardFileReader { std::string path; explicit StandardFileReader(const std::string& p) : path(p) {} }; }
